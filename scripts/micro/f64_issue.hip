// Micro-benchmark: cycles per instruction of one wave64 on one SIMD for the fp64 ops the tile Cholesky is made of.
// Build: hipcc --offload-arch=gfx950 -O2 f64_issue.hip -o f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int MODE>
__global__ void k(double* out, long long* cyc, double x0) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = x0 + i + threadIdx.x;
  double b = x0 * 0.5, c = 1.0 - x0;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    if (MODE == 0) {  // 8 independent FMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = __builtin_fma(a[i], b, c); asm volatile("" : "+v"(a[i])); }
    } else if (MODE == 1) {  // dependent FMA chain
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[0] = __builtin_fma(a[0], b, c); asm volatile("" : "+v"(a[0])); }
    } else if (MODE == 2) {  // readlane pair + FMA with the SGPR operand (independent accumulators)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[(i + 1) & 7]), 5), __builtin_amdgcn_readlane(__double2loint(a[(i + 1) & 7]), 5));
        a[i] = __builtin_fma(s, b, a[i]); asm volatile("" : "+v"(a[i]));
      }
    } else if (MODE == 3) {  // dependent: rcp -> fma -> fma
#pragma unroll
      for (int i = 0; i < 4; ++i) { double r = __builtin_amdgcn_rcp(a[0]); asm volatile("" : "+v"(r)); a[0] = __builtin_fma(r, b, c); asm volatile("" : "+v"(a[0])); }
    } else if (MODE == 4) {  // independent rcp
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = __builtin_amdgcn_rcp(a[i]); asm volatile("" : "+v"(a[i])); }
    } else if (MODE == 5) {  // dependent: fma -> readlane -> fma (pivot broadcast on the chain)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[0]), 5), __builtin_amdgcn_readlane(__double2loint(a[0]), 5));
        a[0] = __builtin_fma(s, b, c); asm volatile("" : "+v"(a[0]));
      }
    } else if (MODE == 6) {  // independent f64 multiply
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = a[i] * b; asm volatile("" : "+v"(a[i])); }
    } else if (MODE == 7) {  // independent FMA with one SGPR-pair operand
      const double s = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(b)), __builtin_amdgcn_readfirstlane(__double2loint(b)));
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = __builtin_fma(s, a[(i + 3) & 7], a[i]); asm volatile("" : "+v"(a[i])); }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; long long* c; hipMalloc(&d, 8 * 64); hipMalloc(&c, 8);
  const char* names[] = {"8 independent v_fma_f64", "dependent v_fma_f64 chain", "readlane pair + fma (indep.)", "dependent rcp -> fma", "independent v_rcp_f64", "dependent fma -> readlane -> fma", "independent v_mul_f64", "independent fma with SGPR operand"};
  const int per[] = {8, 8, 8, 8, 8, 8, 8, 8};
  for (int m = 0; m < 8; ++m) {
    long long h = 0;
    for (int rep = 0; rep < 3; ++rep) {
      switch (m) {
        case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, d, c, 1.000001); break;
        case 7: hipLaunchKernelGGL(k<7>, 1, 64, 0, 0, d, c, 1.000001); break;
      }
      hipDeviceSynchronize(); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    }
    printf("%-36s %8.2f cycles (s_memtime units) per group of %d -> %.2f per op\n", names[m], (double)h / REP, per[m], (double)h / REP / per[m]);
  }
  return 0;
}
