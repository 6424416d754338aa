// Register / LDS building blocks of the tile Cholesky (dense_chol.hip), shared with the single-launch solver of small problems
// (local_lm.hip): the stacked one-wave panel factorisation and the 16 x 16 quadrant products of v_mfma_f64_16x16x4_f64.
#pragma once
#include "common.h"

namespace mpsfm {

constexpr int kTile = 32;
constexpr int kTileElems = kTile * kTile;

__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                          __builtin_amdgcn_readlane(__double2loint(v), l));
}
// 1/sqrt(d): v_rsq_f64 seed + two Newton steps (off the exact-sqrt/divide latency chain)
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  y = y * __builtin_fma(-hd * y, y, 1.5);
  y = y * __builtin_fma(-hd * y, y, 1.5);  // second step kept: v_rsq_f64 alone is good to ~2^-26
  return y;
}

#define MPSFM_PIN(x) asm volatile("" : "+v"(x))

// ---- stacked panel factorisation: Cholesky of the diagonal tile and the solve of the workgroup's own tile in ONE wave --
// Lane i < 32 holds row i of the (updated, symmetric, both triangles valid) diagonal tile D, lane 32 + i holds row i of
// the workgroup's own tile X (or of the identity for the workgroup that owns the diagonal tile).  The column
// operations of a Cholesky panel factorisation of the stacked 64 x 32 matrix [D; X] are the same for both halves —
//   l_iJ = a_iJ / sqrt(d_J),   a_ic -= l_iJ l_cJ  (c > J)
// — so the triangular solve X L^-T costs no instruction of its own and needs no second wave, no hand-shake and no
// LDS traffic per column.
//
// One wave issues one instruction per ~4-5 cycles (fp64 multiply-add: 5.5, v_readlane: 4.9, v_rsq_f64 / v_rcp_f64: 16.5;
// a dependent fp64 op waits 8 — scripts/micro/f64_issue.hip), and the 496 rank-1 multiply-adds per lane are fixed, so
// the factorisation is bound by its INSTRUCTION COUNT: per column one transcendental (v_rsq_f64 + one cubic step
// y = y0 (1 + e/2 + 3e^2/8), e = 1 - d y0^2: v_rsq_f64 is good to ~2^-26, the step to ~e^3), the scaling l = a y, and
// the rank-1 update with the multipliers l_cJ broadcast
//   inside a panel of kSP columns   by v_readlane (they sit in lane c, register J — the matrix is symmetric),
//   beyond the panel                from LDS: after a panel its kSP columns of L (rows < 32) are written once and come
//                                   back as uniform-address ds_read_b128; the next panel's columns get the rank-kSP
//                                   update at once, the columns beyond get it one column of L per column of the next
//                                   panel (filling the issue slots its dependent chain leaves empty).
constexpr int kSP = 8;

// Broadcast reads of L rows from LDS are issued in ONE batch well before their use, as explicit ds_read_b128 with one
// s_waitcnt in front of the consumers: left to itself the compiler issues two or three reads and waits for them
// (s_waitcnt lgkmcnt(0)) ten times per panel boundary, exposing the LDS latency each time (0.45 us per boundary
// measured).  LDS operations of a wave return in order, so the compiler's own counted waits stay correct beside these.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const double* p) { return (uint32_t)(uintptr_t)p; }  // low 32 bits of a __shared__ address = LDS offset
template <int N>
__device__ __forceinline__ void lds_row_load(uint32_t addr, v2d (&m)[N]) {
#pragma unroll
  for (int q = 0; q < N; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m[q]) : "v"(addr), "n"(16 * q));
}
// the loaded values pass through the wait, so no consumer can be scheduled in front of it
__device__ __forceinline__ void lds_wait(v2d (&m)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3])); }
__device__ __forceinline__ void lds_wait(v2d (&m)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]));
}
__device__ __forceinline__ void lds_wait(v2d (&m)[1]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m[0])); }

// END (a multiple of kSP): the columns from END on are padding — identity rows in D, zeros in X — and stay what they are, so the
// factorisation stops there (the last tile column of a small system, local_lm.hip)
template <int J0, int J, int END>
__device__ __forceinline__ void stacked_col(double (&a)[kTile], const double* s_P, bool& ok) {
  constexpr int C = J0 + J;
  constexpr bool kHasFar = (J0 > 0) && (J0 + kSP < END);  // the previous panel still owes the columns beyond this panel
  constexpr int F0 = J0 + kSP, NF = kHasFar ? (END - F0) / 2 : 1;
  // column J of the previous panel for the far columns [F0, 32): requested first, consumed after the chain below
  v2d mf[NF];
  if constexpr (kHasFar) lds_row_load<NF>(lds_addr(s_P + J * kTile + F0), mf);
  const double d = readlane_f64(a[C], C);
  ok = ok && (d > 0.0) && isfinite(d);
  const double y0 = __builtin_amdgcn_rsq(d);
  const double h = d * y0;
  const double e = __builtin_fma(-h, y0, 1.0);
  const double p = __builtin_fma(0.375 * e, e, 0.5 * e);
  const double y = __builtin_fma(y0, p, y0);
  const double l = a[C] * y;
  a[C] = l;
#pragma unroll
  for (int c = J + 1; c < kSP; ++c) a[J0 + c] = __builtin_fma(-l, readlane_f64(l, J0 + c), a[J0 + c]);
  if constexpr (kHasFar) {
    const double lk = a[J0 - kSP + J];
    lds_wait(mf);
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      a[F0 + 2 * q] = __builtin_fma(-lk, mf[q].x, a[F0 + 2 * q]);
      a[F0 + 2 * q + 1] = __builtin_fma(-lk, mf[q].y, a[F0 + 2 * q + 1]);
    }
  }
  if constexpr (J + 1 < kSP) stacked_col<J0, J + 1, END>(a, s_P, ok);
}

template <int J0, int END = kTile>
__device__ __forceinline__ void stacked_panel(double (&a)[kTile], int lane, double* s_P, bool& ok) {
  static_assert(END % kSP == 0 && END <= kTile, "whole panels");
  stacked_col<J0, 0, END>(a, s_P, ok);
  if constexpr (J0 + kSP < END) {
    // columns J0 .. J0+kSP-1 of L, rows < 32: s_P[k][c] = L[c][J0 + k].  The far updates of the previous panel read
    // s_P during this panel: they are all issued by now (one wave, DS operations complete in order).
    __builtin_amdgcn_wave_barrier();
    if (lane < kTile) {
#pragma unroll
      for (int k = 0; k < kSP; ++k) s_P[k * kTile + lane] = a[J0 + k];
    }
    // one wave: its DS operations complete in order, so the reads below see the stores above; only the compiler must
    // be kept from moving them (workgroup-scope fences here cost ~0.3 us per panel boundary)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // rank-kSP update of the next panel's columns (its chain starts on them): all reads first, two batches
    {
      // 4 reads per column of L; two halves of 16 reads (at most 15 LDS operations are in flight, and 32 rows at once
      // would push the wave past its 256 registers)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        v2d m[kSP / 2][kSP / 2];
#pragma unroll
        for (int k = 0; k < kSP / 2; ++k) lds_row_load<kSP / 2>(lds_addr(s_P + (half * (kSP / 2) + k) * kTile + J0 + kSP), m[k]);
#pragma unroll
        for (int k = 0; k < kSP / 2; ++k) lds_wait(m[k]);
#pragma unroll
        for (int k = 0; k < kSP / 2; ++k) {
#pragma unroll
          for (int q = 0; q < kSP / 2; ++q) {
            a[J0 + kSP + 2 * q] = __builtin_fma(-a[J0 + half * (kSP / 2) + k], m[k][q].x, a[J0 + kSP + 2 * q]);
            a[J0 + kSP + 2 * q + 1] = __builtin_fma(-a[J0 + half * (kSP / 2) + k], m[k][q].y, a[J0 + kSP + 2 * q + 1]);
          }
        }
      }
    }
    stacked_panel<J0 + kSP, END>(a, lane, s_P, ok);
  }
}

// ---- 16 x 16 quadrants of a 32 x 32 tile in the accumulator layout of v_mfma_f64_16x16x4_f64 (one per wave) -----------
// element r of lane l sits at row 16 mi + (l >> 4) + 4 r, column 16 ni + (l & 15)
template <typename Ptr>
__device__ __forceinline__ void quad_load(Ptr T, int ld, int lane, int mi, int ni, v4d& acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = T[(16 * mi + (lane >> 4) + 4 * r) * ld + 16 * ni + (lane & 15)];
}
template <typename Ptr>
__device__ __forceinline__ void quad_store(Ptr T, int ld, int lane, int mi, int ni, const v4d& acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) T[(16 * mi + (lane >> 4) + 4 * r) * ld + 16 * ni + (lane & 15)] = acc[r];
}
// operands of a quadrant product over k = 0..31: lane l takes row 16 b + (l & 15), columns 8 (l >> 4) .. + 7 of a row-major
// tile (the k index is permuted identically for both operands, which leaves the sum unchanged: 64 contiguous bytes per lane)
__device__ __forceinline__ void quad_operand(const double* __restrict__ T, int lane, int b, double (&o)[8]) {
  const double2* p = reinterpret_cast<const double2*>(T + (16 * b + (lane & 15)) * kTile + 8 * (lane >> 4));
#pragma unroll
  for (int s = 0; s < 4; ++s) { const double2 x = p[s]; o[2 * s] = x.x; o[2 * s + 1] = x.y; }
}
// acc -= A[16 mi .., :] B[16 ni .., :]^T
__device__ __forceinline__ void quad_gemm_sub(const double (&a)[8], const double (&b)[8], v4d& acc) {
#pragma unroll
  for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[s], b[s], acc, 0, 0, 0);
}

}  // namespace mpsfm
