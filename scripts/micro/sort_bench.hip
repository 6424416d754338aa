// How long rocPRIM's sorts take at the landmark-order size (150 k pairs): hipcc --offload-arch=gfx950 -O3 sort_bench.hip -o sort_bench
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_merge_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
struct KP { unsigned long long k; int p; int pad; };
struct LessKP { __host__ __device__ bool operator()(const KP& a, const KP& b) const { return a.k != b.k ? a.k < b.k : a.p < b.p; } };
struct Less { __host__ __device__ bool operator()(unsigned long long a, unsigned long long b) const { return a < b; } };
int main() {
  for (int n : {4000, 150000, 800000}) {
    std::vector<unsigned long long> k(n); std::vector<int> v(n);
    std::mt19937_64 g(1);
    // keys shaped like the landmark order's: class 2 bits (all zero) | six 8-bit slots (neighbouring cameras, 255 = none) | 9-bit count
    for (int i = 0; i < n; ++i) {
      const int c0 = g() % 199, len = 2 + g() % 6;
      unsigned long long key = 0;
      for (int q = 0; q < 6; ++q) key = (key << 8) | (unsigned long long)(q < len ? (c0 + q) % 199 : 255);
      k[i] = (key << 9) | (unsigned long long)len; v[i] = i;
    }
    unsigned long long *dk, *dk2; int *dv, *dv2; unsigned *dk32, *dk32b;
    hipMalloc(&dk, 8 * n); hipMalloc(&dk2, 8 * n); hipMalloc(&dv, 4 * n); hipMalloc(&dv2, 4 * n); hipMalloc(&dk32, 4 * n); hipMalloc(&dk32b, 4 * n);
    hipMemcpy(dk, k.data(), 8 * n, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(dk32, k.data(), 4 * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void* tmp = nullptr; size_t tb = 0;
    auto time = [&](const char* what, auto&& f) {
      tb = 0; f(nullptr); hipFree(tmp); hipMalloc(&tmp, tb + 16);
      f(tmp); hipDeviceSynchronize();
      float best = 1e9;
      for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(tmp); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
      printf("n %7d  %-34s %8.3f ms (temp %zu B)\n", n, what, best, tb);
    };
    time("radix pairs u64 bits 0..64", [&](void* t) { rocprim::radix_sort_pairs(t, tb, dk, dk2, dv, dv2, (size_t)n, 0, 64); });
    time("radix pairs u64 bits 0..59", [&](void* t) { rocprim::radix_sort_pairs(t, tb, dk, dk2, dv, dv2, (size_t)n, 0, 59); });
    time("radix pairs u64 bits 0..32", [&](void* t) { rocprim::radix_sort_pairs(t, tb, dk, dk2, dv, dv2, (size_t)n, 0, 32); });
    time("radix pairs u32 bits 0..32", [&](void* t) { rocprim::radix_sort_pairs(t, tb, dk32, dk32b, dv, dv2, (size_t)n, 0, 32); });
    time("merge_sort pairs u64", [&](void* t) { rocprim::merge_sort(t, tb, dk, dk2, dv, dv2, (size_t)n, Less()); });
    std::vector<KP> kp(n); for (int i = 0; i < n; ++i) kp[i] = KP{k[i], i, 0};
    KP *dkp, *dkp2; hipMalloc(&dkp, 16 * n); hipMalloc(&dkp2, 16 * n); hipMemcpy(dkp, kp.data(), 16 * n, hipMemcpyHostToDevice);
    time("merge_sort keys {u64, int}", [&](void* t) { rocprim::merge_sort(t, tb, dkp, dkp2, (size_t)n, LessKP()); });
  }
  return 0;
}
