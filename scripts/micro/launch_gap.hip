// Micro-benchmark: time per launch of a chain of tiny dependent kernels in one stream, plain launches vs a
// captured graph.  Build: hipcc --offload-arch=gfx950 -O2 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_tiny(double* p, int j) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += (double)j; }
__global__ void k_mid(double* p, int j) {  // ~700 workgroups of 128 threads touching 8 KB each, like a Cholesky step's grid
  const size_t o = (size_t)blockIdx.x * 1024 + threadIdx.x * 8;
  double s = 0; for (int k = 0; k < 8; ++k) s += p[o + k];
  p[o] = s * 0.5 + j;
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 1024 * 1024); hipMemset(d, 0, sizeof(double) * 1024 * 1024);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 38, R = 50;
  for (int variant = 0; variant < 2; ++variant) {
    auto chain = [&]() { for (int j = 0; j < N; ++j) { if (variant == 0) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d, j); else hipLaunchKernelGGL(k_mid, dim3(700), dim3(128), 0, s, d, j); } };
    chain(); hipStreamSynchronize(s);
    hipEventRecord(e0, s); for (int r = 0; r < R; ++r) chain(); hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d plain: %.2f us per launch\n", variant, 1e3 * ms / (N * R));
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); for (int r = 0; r < R; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d graph: %.2f us per kernel node\n", variant, 1e3 * ms / (N * R));
    // one chain with a host sync after each (what the LM loop sees per iteration)
    hipEventRecord(e0, s); for (int r = 0; r < R; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); } hipEventRecord(e1, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d graph + sync per chain: %.2f us per kernel node\n", variant, 1e3 * ms / (N * R));
  }
  return 0;
}
