// Micro-benchmark: what a hand-off between workgroups INSIDE one launch costs, against the ~3.3 us of a launch boundary.
//   1. ping-pong of an 8 KB tile + a flag between two workgroups (different XCD: blocks 0/1, same XCD: blocks 0/8)
//   2. a grid-wide barrier over G co-resident workgroups (atomic counter + bounded spin)
// Every spin is bounded (kSpinMax polls), a miss raises a flag and every workgroup leaves.
// Build: hipcc --offload-arch=gfx950 -O2 flag_latency.hip -o flag_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kSpinMax = 1 << 22;

__device__ inline int ld_acq(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_rel(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline bool wait_ge(const int* p, int v, int* fail) {
  for (int i = 0; i < kSpinMax; ++i) {
    if (ld_acq(p) >= v) return true;
    if ((i & 1023) == 1023 && ld_acq(fail)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  st_rel(fail, 1);
  return false;
}

// blocks `a` and `b` play; everyone else leaves at once.  256 threads, each moves 4 doubles = 8 KB per hand-off.
__global__ void __launch_bounds__(256) k_pingpong(double* tile, int* flags, int* fail, int a, int b, int rounds, int with_tile,
                                                  long long* cycles) {
  const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (me < 0) return;
  __shared__ int ok;
  double acc[4] = {1.0, 2.0, 3.0, 4.0};
  double* mine = tile + me * 1024 + threadIdx.x * 4;
  const double* theirs = tile + (1 - me) * 1024 + threadIdx.x * 4;
  const long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    if (me == 0) {
      if (with_tile) for (int k = 0; k < 4; ++k) mine[k] = acc[k] + r;
      __syncthreads();
      if (threadIdx.x == 0) { __threadfence(); st_rel(flags + 0, r); }
    }
    if (threadIdx.x == 0) ok = wait_ge(flags + (1 - me), r, fail) ? 1 : 0;
    __syncthreads();
    if (!ok) return;
    if (with_tile) for (int k = 0; k < 4; ++k) acc[k] += theirs[k];
    if (me == 1) {
      if (with_tile) for (int k = 0; k < 4; ++k) mine[k] = acc[k] - r;
      __syncthreads();
      if (threadIdx.x == 0) { __threadfence(); st_rel(flags + 1, r); }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && me == 0) cycles[0] = wall_clock64() - t0;
  if (acc[0] == 123.456) tile[4000] = acc[1] + acc[2] + acc[3];
}

__global__ void __launch_bounds__(256) k_gridbar(int* counter, int* fail, int rounds, long long* cycles, double* sink) {
  __shared__ int ok;
  const long long t0 = wall_clock64();
  double v = threadIdx.x;
  for (int r = 1; r <= rounds; ++r) {
    v = v * 1.0000001 + 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      ok = wait_ge(counter, r * (int)gridDim.x, fail) ? 1 : 0;
    }
    __syncthreads();
    if (!ok) return;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = wall_clock64() - t0;
  if (v == 123.456) sink[0] = v;
}

int main() {
  double* tile; int* flags; int* fail; long long* cyc; int* counter;
  hipMalloc(&tile, sizeof(double) * 8192); hipMalloc(&flags, 64); hipMalloc(&fail, 64); hipMalloc(&cyc, 64); hipMalloc(&counter, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int R = 2000;
  int clk_khz = 100000;  // wall_clock64 ticks at 100 MHz on gfx9
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
  const int pairs[3][2] = {{0, 1}, {0, 8}, {0, 3}};
  for (int with_tile = 0; with_tile < 2; ++with_tile)
    for (int p = 0; p < 3; ++p) {
      hipMemsetAsync(flags, 0, 64, s); hipMemsetAsync(fail, 0, 64, s); hipMemsetAsync(tile, 0, sizeof(double) * 8192, s);
      hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(256), 0, s, tile, flags, fail, pairs[p][0], pairs[p][1], R, with_tile, cyc);
      hipStreamSynchronize(s);
      long long c; int f; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
      printf("ping-pong blocks %d<->%d %s: %.3f us per one-way hand-off%s\n", pairs[p][0], pairs[p][1], with_tile ? "flag + 8 KB tile" : "flag only",
             1e3 * (double)c / clk_khz / (2.0 * R), f ? "  [SPIN BOUND HIT]" : "");
    }
  for (int G : {8, 32, 64, 128, 256}) {
    hipMemsetAsync(counter, 0, 64, s); hipMemsetAsync(fail, 0, 64, s);
    hipLaunchKernelGGL(k_gridbar, dim3(G), dim3(256), 0, s, counter, fail, R, cyc, tile);
    hipStreamSynchronize(s);
    long long c; int f; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("grid barrier over %3d workgroups: %.3f us per barrier%s\n", G, 1e3 * (double)c / clk_khz / R, f ? "  [SPIN BOUND HIT]" : "");
  }
  return 0;
}
