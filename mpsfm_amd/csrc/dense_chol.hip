// Dense solve of the reduced camera system (gfx950): what Ceres' SPARSE_SCHUR hands to CHOLMOD
// (reference call site mpsfm/sfm/mapper/bundle_adjustment.py:285-293), done here as a tiled
// right-looking Cholesky on fp64 MFMA (v_mfma_f64_16x16x4_f64).
//
// Storage: lower triangle of 32x32 tiles, tile (ti, tj) at lt_tile(ti, tj) * 1024 doubles,
// row-major inside a tile.  Tile row `nt` (one extra) carries the right-hand side in its row 0, so
// the forward substitution z = L^-1 rhs falls out of the factorisation; k_backsub then solves
// L^T y = z.
#include "common.h"

namespace mpsfm {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kTile = 32;
constexpr int kTileElems = kTile * kTile;
constexpr int kPlainMaxTiles = 64;  // up to this many tile columns: one outer panel, inverse propagation instead of back substitution

// one workgroup per lower tile (including the rhs tile row)
__global__ __launch_bounds__(256) void k_assemble(AssembleArgs P) {
  // decode tile id -> (ti, tj), ti >= tj, ti in [0, nt], tj in [0, nt-1]
  const int64_t id = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
  while ((int64_t)ti * (ti + 1) / 2 > id) --ti;
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= id) ++ti;
  const int tj = (int)(id - (int64_t)ti * (ti + 1) / 2);
  double* T = P.A + id * kTileElems;
  const int n = P.n;
  // the accumulators of the inverse propagation (see k_chol_step) start from zero
  if (P.Pinv && ti < P.nt && tj < ti)
    for (int e = threadIdx.x; e < kTileElems; e += 256) P.Pinv[id * kTileElems + e] = 0.0;
  for (int e = threadIdx.x; e < kTileElems; e += 256) {
    const int r = e >> 5, c = e & 31;
    const int C = tj * kTile + c;
    double v = 0.0;
    if (ti == P.nt) {
      if (tj < P.nt && r == 0 && C < n) v = P.wv[C] - P.gc[C];
    } else {
      const int R = ti * kTile + r;
      if (R >= n || C >= n) {
        v = (R == C) ? 1.0 : 0.0;
      } else {
        const int br = R / 6, a = R - br * 6, bc = C / 6, b = C - bc * 6;
        if (br < bc) v = P.Sblk[ut_block(br, bc, P.ncv) * 36 + a * 6 + b];
        else if (br > bc) v = P.Sblk[ut_block(bc, br, P.ncv) * 36 + b * 6 + a];
        else v = P.Sblk[ut_block(br, br, P.ncv) * 36 + (a <= b ? a * 6 + b : b * 6 + a)];
        if (R == C) v += fmin(fmax(P.diagU[R], P.min_diag), P.max_diag) / P.radius;
      }
    }
    T[e] = v;
  }
}

// C(32x32, C-layout accumulators) -= A(32x32) * B(32x32)^T, A/B row-major tiles in global memory.
// Lane l supplies A[16 mi + (l&15)][8 (l>>4) + s] at k-step s: the k index is permuted identically
// for both operands, which leaves the sum unchanged and lets each lane read 64 contiguous bytes.
__device__ __forceinline__ void tile_syrk_sub(const double* __restrict__ At, const double* __restrict__ Bt, int lane,
                                              v4d acc[2][2]) {
  const int row = lane & 15, kg = lane >> 4;
  double a[2][8], b[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double2* pa = reinterpret_cast<const double2*>(At + (16 * h + row) * kTile + 8 * kg);
    const double2* pb = reinterpret_cast<const double2*>(Bt + (16 * h + row) * kTile + 8 * kg);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double2 x = pa[s], y = pb[s];
      a[h][2 * s] = -x.x; a[h][2 * s + 1] = -x.y;
      b[h][2 * s] = y.x;  b[h][2 * s + 1] = y.y;
    }
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
}

// C-layout element (mi, ni, reg r) of lane l sits at row 16 mi + (l>>4) + 4 r, col 16 ni + (l&15)
__device__ __forceinline__ void tile_load_acc(const double* __restrict__ T, int lane, v4d acc[2][2]) {
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[mi][ni][r] = T[(16 * mi + (lane >> 4) + 4 * r) * kTile + 16 * ni + (lane & 15)];
}
template <typename Ptr>
__device__ __forceinline__ void tile_store_acc(Ptr T, int ld, int lane, const v4d acc[2][2]) {
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        T[(16 * mi + (lane >> 4) + 4 * r) * ld + 16 * ni + (lane & 15)] = acc[mi][ni][r];
}

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

// Cholesky of a 32x32 tile held one row per lane (lane i and lane i+32 both hold row i).
// Right-looking, software-pipelined by one column so that the only cross-lane traffic on the
// dependent chain is two readlane broadcasts:
//   chain(j)  d = a_jj (readlane), inv = rsqrt(d), t = a_ij / d, the NEXT pivot column gets its
//             update right away (a[j+1] -= t * a_{j+1,j}, readlane), column j (unscaled) goes to LDS
//   bulk(j-1) a[c] -= t_{j-1} * a_{c,j-1} for c >= j+1 with the column broadcast from LDS; its
//             reads are issued before chain(j) and consumed after it, hiding the LDS round trip.
// On return lane i holds row i of L in a[0..i]; s_inv[j] = 1 / L[j][j].
#define MPSFM_PIN(x) asm volatile("" : "+v"(x))

constexpr int kPanel = 16;  // columns factored per register panel; the rest of the tile is updated by MFMA

// bulk(J-1) slot SLOT: up to Q multiply-adds a[c] -= tprev * col[c], c = J+1+SLOT*Q ..
template <int J, int SLOT, int Q, int NF>
__device__ __forceinline__ void potrf_bulk(double (&a)[kPanel], const double (&col)[kPanel], double tprev) {
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    constexpr int kBase = J + 1 + SLOT * Q;
    if (SLOT * Q + u < NF) {
      a[kBase + u] -= tprev * col[kBase + u];
      MPSFM_PIN(a[kBase + u]);
    }
  }
}

// Column J of a 16-column register panel (tile columns OFF .. OFF+15; lane i holds row i of the tile,
// a[c] = A[i][OFF + c]).  Right-looking inside the panel, software-pipelined by one column:
//   chain(J)  d = pivot (readlane), 1/d by v_rcp_f64 + Newton, t = a_iJ / d, the NEXT pivot column gets
//             its update right away (readlane), the unscaled column goes to LDS
//   bulk(J-1) a[c] -= t_{J-1} * a_{OFF+c, J-1} for the remaining panel columns, column broadcast from LDS
// Source order == issue order (values are pinned with empty volatile asm statements): the dependent
// ops of chain(J) are interleaved with the independent multiply-adds of bulk(J-1).
// Publishes column OFF+J of L (s_Lt), 1/L[jj][jj] (s_inv) and the progress counter for the TRSM wave.
template <int J, int OFF>
__device__ __forceinline__ void potrf_panel_col(double (&a)[kPanel], int lane, double* s_inv, double (*s_col)[kTile],
                                                double* s_Lt, int* s_ready, double& tprev, bool& ok) {
  constexpr int NF = (J >= 1) ? (kPanel - 1 - J) : 0;
  constexpr int Q = (NF + 9) / 10;
  double col[kPanel];
  if (J >= 1) {
#pragma unroll
    for (int c = J + 1; c < kPanel; ++c) col[c] = s_col[(J - 1) & 1][OFF + c];
  }
  const double d = readlane_f64(a[J], OFF + J);
  ok = ok && (d > 0.0) && isfinite(d);
  double r = __builtin_amdgcn_rcp(d); MPSFM_PIN(r);
  double y = __builtin_amdgcn_rsq(d); MPSFM_PIN(y);          // off-chain: 1/sqrt(d) for L itself
  potrf_bulk<J, 0, Q, NF>(a, col, tprev);
  double e = __builtin_fma(-d, r, 1.0); MPSFM_PIN(e);
  double hd = 0.5 * d; MPSFM_PIN(hd);
  potrf_bulk<J, 1, Q, NF>(a, col, tprev);
  r = __builtin_fma(r, e, r); MPSFM_PIN(r);
  double w = -hd * y; MPSFM_PIN(w);
  potrf_bulk<J, 2, Q, NF>(a, col, tprev);
  e = __builtin_fma(-d, r, 1.0); MPSFM_PIN(e);                // second step: v_rcp_f64 alone is ~2^-26
  double f = __builtin_fma(w, y, 1.5); MPSFM_PIN(f);
  potrf_bulk<J, 3, Q, NF>(a, col, tprev);
  r = __builtin_fma(r, e, r); MPSFM_PIN(r);
  y = y * f; MPSFM_PIN(y);
  potrf_bulk<J, 4, Q, NF>(a, col, tprev);
  double t = a[J] * r; MPSFM_PIN(t);                          // a_iJ / d
  w = -hd * y; MPSFM_PIN(w);
  potrf_bulk<J, 5, Q, NF>(a, col, tprev);
  if (J + 1 < kPanel) {
    a[(J + 1) % kPanel] -= t * readlane_f64(a[J], OFF + (J + 1) % kPanel);
    MPSFM_PIN(a[(J + 1) % kPanel]);
  }
  f = __builtin_fma(w, y, 1.5); MPSFM_PIN(f);
  potrf_bulk<J, 6, Q, NF>(a, col, tprev);
  const double inv = y * f;
  potrf_bulk<J, 7, Q, NF>(a, col, tprev);
  const double l = a[J] * inv;                                // l_iJ = a_iJ / sqrt(d)
  potrf_bulk<J, 8, Q, NF>(a, col, tprev);
  if (lane < kTile) { s_col[J & 1][lane] = a[J]; s_Lt[(OFF + J) * kTile + lane] = l; }
  if (lane == 0) {
    s_inv[OFF + J] = inv;
    // column OFF+J of L and its 1/diag are in LDS: the TRSM wave may use them (DS ops of a wave complete in order)
    __hip_atomic_store(s_ready, OFF + J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  a[J] = l;
  potrf_bulk<J, 9, Q, NF>(a, col, tprev);
  tprev = t;
  if constexpr (J + 1 < kPanel) potrf_panel_col<J + 1, OFF>(a, lane, s_inv, s_col, s_Lt, s_ready, tprev, ok);
}

// Blocked Cholesky of the 32x32 tile in s_T (row stride 33, both triangles valid) by one wave:
//   panel 0  columns 0..15 of all 32 rows in registers (factors A11 and solves A21 in one go)
//   update   A22 -= L21 L21^T with four v_mfma_f64_16x16x4_f64 (operands read back from s_Lt)
//   panel 1  columns 16..31 (rows 16..31 matter)
// L^T ends up in s_Lt (s_Lt[c*32 + r] = L[r][c], r >= c), 1/diag in s_inv.
__device__ __forceinline__ bool potrf_tile(double (*s_T)[kTile + 1], int lane, double* s_inv, double (*s_col)[kTile], double* s_Lt,
                                           int* s_ready) {
  bool ok = true;
  const int row = lane & 31;
  {
    double a[kPanel];
#pragma unroll
    for (int c = 0; c < kPanel; ++c) a[c] = s_T[row][c];
    double tprev = 0.0;
    potrf_panel_col<0, 0>(a, lane, s_inv, s_col, s_Lt, s_ready, tprev, ok);
  }
  {
    const int i = lane & 15, kg = lane >> 4;
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = s_T[kPanel + kg + 4 * r][kPanel + i];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double op = s_Lt[(4 * s + kg) * kTile + kPanel + i];  // L21[i][4s + kg]
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-op, op, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_T[kPanel + kg + 4 * r][kPanel + i] = acc[r];
  }
  {
    double a[kPanel];
#pragma unroll
    for (int c = 0; c < kPanel; ++c) a[c] = s_T[row][kPanel + c];
    double tprev = 0.0;
    potrf_panel_col<0, kPanel>(a, lane, s_inv, s_col, s_Lt, s_ready, tprev, ok);
  }
  return ok;
}

// x <- x L^-T for the row held by this lane, columns [C0, C1): L^T is read from LDS as
// s_Lt[c*32 + k] = L[k][c] (uniform addresses: broadcast reads), 1/L[c][c] from s_inv.  The reads of
// column c+1 are issued before the multiply-adds of column c.
template <int C0, int C1>
__device__ __forceinline__ void trsm_cols(double (&x)[kTile], const double* s_Lt, const double* s_inv) {
  double lt[2][kTile];
  double iv[2];
#pragma unroll
  for (int k = C0 + 1; k < kTile; ++k) lt[C0 & 1][k] = s_Lt[C0 * kTile + k];
  iv[C0 & 1] = s_inv[C0];
#pragma unroll
  for (int c = C0; c < C1; ++c) {
    if (c + 1 < C1) {
#pragma unroll
      for (int k = c + 2; k < kTile; ++k) lt[(c + 1) & 1][k] = s_Lt[(c + 1) * kTile + k];
      iv[(c + 1) & 1] = s_inv[c + 1];
    }
    x[c] *= iv[c & 1];
#pragma unroll
    for (int k = c + 1; k < kTile; ++k) {
      x[k] -= x[c] * lt[c & 1][k];
      asm volatile("" : "+v"(x[k]));
    }
  }
}

// The solve runs BEHIND the factorising wave: the first 16 columns (76 % of the multiply-adds) start
// as soon as the progress counter says columns 0..15 of L are in LDS, i.e. under the second half of
// the factorisation; only the last 16 columns wait for the factor to be complete.
__device__ __forceinline__ void trsm_row(double (&x)[kTile], const double* s_Lt, const double* s_inv, int* s_ready) {
  while (__hip_atomic_load(s_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kTile / 2) __builtin_amdgcn_s_sleep(2);
  trsm_cols<0, kTile / 2>(x, s_Lt, s_inv);
  while (__hip_atomic_load(s_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kTile) __builtin_amdgcn_s_sleep(2);
  trsm_cols<kTile / 2, kTile>(x, s_Lt, s_inv);
}

// half-tile variants (rows 16 mi .. 16 mi + 15) for the two waves of a trailing-update workgroup
__device__ __forceinline__ void half_load_acc(const double* __restrict__ T, int lane, int mi, v4d acc[2]) {
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[ni][r] = T[(16 * mi + (lane >> 4) + 4 * r) * kTile + 16 * ni + (lane & 15)];
}
__device__ __forceinline__ void half_store_acc(double* __restrict__ T, int lane, int mi, const v4d acc[2]) {
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(16 * mi + (lane >> 4) + 4 * r) * kTile + 16 * ni + (lane & 15)] = acc[ni][r];
}
__device__ __forceinline__ void half_syrk_sub(const double* __restrict__ At, const double* __restrict__ Bt, int lane, int mi,
                                              v4d acc[2]) {
  const int row = lane & 15, kg = lane >> 4;
  double a[8], b[2][8];
  {
    const double2* pa = reinterpret_cast<const double2*>(At + (16 * mi + row) * kTile + 8 * kg);
#pragma unroll
    for (int s = 0; s < 4; ++s) { const double2 x = pa[s]; a[2 * s] = -x.x; a[2 * s + 1] = -x.y; }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double2* pb = reinterpret_cast<const double2*>(Bt + (16 * h + row) * kTile + 8 * kg);
#pragma unroll
    for (int s = 0; s < 4; ++s) { const double2 y = pb[s]; b[h][2 * s] = y.x; b[h][2 * s + 1] = y.y; }
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[ni][s], acc[ni], 0, 0, 0);
}

// ---- trailing update of an outer panel: 64x64 output block per workgroup ---------------------------
// C(ti, tk) -= sum_{c = c0..j} L(ti, c) L(tk, c)^T for the tiles right of the panel.  One wave per 32x32 output
// tile of a 2x2 tile block; the four operand tiles of a panel column (two row tiles, two column tiles) are
// staged in LDS once per workgroup and double-buffered against the MFMAs, so every operand byte fetched from
// L2 feeds 8 flops instead of 2.7 (the per-tile kernel is L2-bandwidth-bound at ~25 % of the MFMA peak).
constexpr int kLdsLd = 34;  // LDS row stride of a staged tile (doubles): 16-byte aligned rows, banks spread
template <int LD>
__device__ __forceinline__ void tile_syrk_sub_lds(const double* At, const double* Bt, int lane, v4d acc[2][2]) {
  const int row = lane & 15, kg = lane >> 4;
  double a[2][8], b[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double2* pa = reinterpret_cast<const double2*>(At + (16 * h + row) * LD + 8 * kg);
    const double2* pb = reinterpret_cast<const double2*>(Bt + (16 * h + row) * LD + 8 * kg);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double2 x = pa[s], y = pb[s];
      a[h][2 * s] = -x.x; a[h][2 * s + 1] = -x.y;
      b[h][2 * s] = y.x;  b[h][2 * s + 1] = y.y;
    }
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
}

// Columns [tk0, tk_hi] of the trailing matrix (all rows ti >= tk up to the rhs row nt) receive panel columns c0..j.
__global__ __launch_bounds__(256) void k_big_update(double* A, int nt, int j, int c0, int tk0, int tk_hi) {
  __shared__ __attribute__((aligned(16))) double s_t[2][4][kTile * kLdsLd];
  const int bx = blockIdx.x, by = blockIdx.y;
  if (bx < by) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti0 = tk0 + 2 * bx, tkb = tk0 + 2 * by;
  const int ti = ti0 + (wave & 1), tk = tkb + (wave >> 1);
  const bool valid = (ti <= nt) && (tk <= tk_hi) && (ti >= tk);
  // staging: thread -> (tile t, 16-byte piece q); 4 tiles x 512 pieces, 8 pieces per thread
  double2 pre[8];
  auto fetch = [&](int c) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + 256 * u, t = idx >> 9, q = idx & 511;
      const int trow = (t < 2) ? ti0 + t : tkb + (t - 2);
      const bool ok = (t < 2) ? (trow <= nt) : (trow <= tk_hi);
      pre[u] = ok ? reinterpret_cast<const double2*>(A + lt_tile(trow, c) * kTileElems)[q] : make_double2(0.0, 0.0);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + 256 * u, t = idx >> 9, q = idx & 511;
      const int r = q >> 4, c2 = q & 15;  // row, double2 column
      *reinterpret_cast<double2*>(&s_t[buf][t][r * kLdsLd + 2 * c2]) = pre[u];
    }
  };
  v4d acc[2][2];
  double* C = A + lt_tile(valid ? ti : tk0, valid ? tk : tk0) * kTileElems;
  if (valid) tile_load_acc(C, lane, acc);
  fetch(c0);
  park(0);
  __syncthreads();
  for (int c = c0; c <= j; ++c) {
    const int buf = (c - c0) & 1;
    if (c < j) fetch(c + 1);
    if (valid) tile_syrk_sub_lds<kLdsLd>(s_t[buf][wave & 1], s_t[buf][2 + (wave >> 1)], lane, acc);
    if (c < j) park(buf ^ 1);
    __syncthreads();
  }
  if (valid) tile_store_acc(C, kTile, lane, acc);
}

int g_dbg_flags = 0;
extern "C" void mpsfm_debug_set(int f) { g_dbg_flags = f; }

// One step of the right-looking factorisation, two waves per tile.  j = -1: factor tile column 0 only.
//   trailing tile (tk > j+1):  A[ti][tk] -= L[ti][j] L[tk][j]^T, one 16-row half per wave.
//   panel tile (tk == j+1):    wave 0 re-derives the updated diagonal tile and factors it in registers,
//                              wave 1 updates the workgroup's own tile meanwhile and then solves it
//                              against the factor (X L^-T); the workgroup that owns the diagonal tile
//                              stores L and L^-T (kept for the back substitution).
// Outer panels (large matrices): c0 is the first tile column whose L is applied to a trailing tile in this
// launch (c0 == j: the plain right-looking step), tk_max the last tile column this launch touches, and
//   kStepNoOwnUpdate  the panel column j+1 has already received column j (first step of an outer panel),
//   kStepBig          no factorisation: every tile (ti, tk), j < tk <= tk_max, gets columns c0..j at once
//                     (one load and one store of the tile for a rank-32*(j-c0+1) update).
constexpr int kStepNoOwnUpdate = 1, kStepBig = 2;
// Inverse propagation (plain path only, Pinv != NULL).  The back substitution y = L^-T z is a chain of nt dependent
// tile solves — ten launches of ~10 us at nt = 38.  Instead the steps also build, in the shadow of their
// latency-bound panel factorisation, the accumulators  P(i,k) = sum_{j=k}^{i-1} L(i,j) X(j,k)  of the inverse
// X = L^-1  (X(k,k) = L(k,k)^-1, X(i,k) = -L(i,i)^-1 P(i,k)), after which
//   y_k = w_k - sum_{i>k} P(i,k)^T w_i,   w_i = L(i,i)^-T z_i
// is two launches.  Launch j uses column j of L and L(j,j)^-1, both final since launch j-1:
//   role (i, k), j < i < nt, k <= j:   P(i,k) += L(i,j) X(j,k),  X(j,k) recomputed from P(j,k) (final: it last
//   changed in launch j-1), so no workgroup reads what another one writes in the same launch.
// Roles are the blocks with blockIdx.y >= cols.
#ifndef MPSFM_INV_ROWS
#define MPSFM_INV_ROWS 2
#endif
constexpr int kInvRows = MPSFM_INV_ROWS;  // rows i handled by one inverse-role workgroup (X(j,k) is formed once for all of them)
__device__ __forceinline__ void inv_role(const double* A, const double* LinvT, double* Pinv, int nt, int j, int i0, int i1, int k,
                                         double (*s_A)[kTile + 1], double (*s_B)[kTile + 1], double* s_C, double (*s_M)[kTile + 1]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* Li = LinvT + (size_t)j * kTileElems;               // L(j,j)^-T, row-major
  const double* Pjk = Pinv + lt_tile(j, k) * kTileElems;           // only read when k < j
  // the first row's accumulator and A operand are requested now: their latency runs under the formation of T
  const int m16p = lane & 15, kgp = lane >> 4;
  v4d acc0[2];
  double a0[8];
  {
    half_load_acc(Pinv + lt_tile(i0, k) * kTileElems, lane, wave, acc0);
    const double* arow = A + lt_tile(i0, j) * kTileElems + (16 * wave + m16p) * kTile + kgp;
#pragma unroll
    for (int s = 0; s < 8; ++s) a0[s] = arow[4 * s];
  }
  for (int e = tid; e < kTileElems; e += 128) {
    const int r = e >> 5, c = e & 31;
    s_B[r][c] = Li[e];
    s_C[e] = (k < j) ? Pjk[e] : 0.0;
  }
  __syncthreads();
  const int m16 = lane & 15, kg = lane >> 4;
  // T = X(j,k): k < j: -(L(j,j)^-1 P(j,k)),  T[m][n] = -sum_q LinvT[q][m] Pjk[q][n];  k == j: L(j,j)^-1, T[m][n] = LinvT[n][m]
  if (k < j) {
    v4d t[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double a = -s_B[4 * s + kg][16 * wave + m16];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) t[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, s_C[(4 * s + kg) * kTile + 16 * ni + m16], t[ni], 0, 0, 0);
    }
    __syncthreads();  // T overwrites L(j,j)^-T in LDS once both waves have consumed it
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_M[16 * wave + kg + 4 * r][16 * ni + m16] = t[ni][r];
  } else {
    double tr[kTileElems / 128];
#pragma unroll
    for (int u = 0; u < kTileElems / 128; ++u) { const int e = tid + 128 * u; tr[u] = s_B[e & 31][e >> 5]; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTileElems / 128; ++u) { const int e = tid + 128 * u; s_M[e >> 5][e & 31] = tr[u]; }
  }
  __syncthreads();
  // P(i,k) += L(i,j) T for the rows of this workgroup; the B operand (T) stays in registers
  double bT[8][2];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bT[s][ni] = s_M[4 * s + kg][16 * ni + m16];
  for (int i = i0; i < i1; ++i) {
    double* Pik = Pinv + lt_tile(i, k) * kTileElems;
    v4d acc[2];
    double a[8];
    if (i == i0) {
      acc[0] = acc0[0]; acc[1] = acc0[1];
#pragma unroll
      for (int s = 0; s < 8; ++s) a[s] = a0[s];
    } else {
      half_load_acc(Pik, lane, wave, acc);
      // A operand straight from the tile in memory: lane (m, q) reads L(i,j)[16 wave + m][4 s + kg]
      const double* arow = A + lt_tile(i, j) * kTileElems + (16 * wave + m16) * kTile + kg;
#pragma unroll
      for (int s = 0; s < 8; ++s) a[s] = arow[4 * s];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bT[s][ni], acc[ni], 0, 0, 0);
    half_store_acc(Pik, lane, wave, acc);
  }
}

__global__ __launch_bounds__(128) void k_chol_step(double* A, double* LinvT, int nt, int j, int* fail, int dbg, int c0, int tk_max,
                                                   int mode, double* Pinv, int cols) {
  __shared__ double s_T[kTile][kTile + 1];
  __shared__ double s_X[kTile][kTile + 1];
  __shared__ double s_Lt[kTile * kTile];
  __shared__ double s_inv[kTile];
  __shared__ double s_col[2][kTile];
  __shared__ int s_ready;  // columns of L published by the factorising wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (Pinv && (int)blockIdx.y >= cols) {
    const int nrows = nt - 1 - j;  // i in (j, nt-1]
    const int ngrp = (nrows + kInvRows - 1) / kInvRows;
    const int role = ((int)blockIdx.y - cols) * (int)gridDim.x + (int)blockIdx.x;
    if (j < 0 || nrows <= 0 || role >= ngrp * (j + 1) || (dbg & 8)) return;  // dbg 8: dispatch the roles, do nothing
    const int i0 = j + 1 + (role % ngrp) * kInvRows;
    inv_role(A, LinvT, Pinv, nt, j, i0, min(i0 + kInvRows, nt), role / ngrp, s_T, s_X, s_Lt, s_X);
    return;
  }
  const int tk = j + 1 + blockIdx.y;
  const int ti = j + 1 + blockIdx.x;
  if (ti < tk || tk >= nt || ti > nt || tk > tk_max) return;
  double* C = A + lt_tile(ti, tk) * kTileElems;
  if (tk != j + 1 || (mode & kStepBig)) {
    v4d acc[2];
    half_load_acc(C, lane, wave, acc);
    if (!(dbg & 4))
      for (int c = c0; c <= j; ++c) half_syrk_sub(A + lt_tile(ti, c) * kTileElems, A + lt_tile(tk, c) * kTileElems, lane, wave, acc);
    half_store_acc(C, lane, wave, acc);
    return;
  }
  const bool own_update = (j >= 0) && !(mode & kStepNoOwnUpdate) && !(dbg & 4);
  // ---- panel column j+1 -------------------------------------------------------------------

  const int row = lane & 31;
  const bool diag = (ti == tk);
  if (threadIdx.x == 0) s_ready = 0;
  if (wave == 0) {
    // updated diagonal tile -> s_T
    v4d dacc[2][2];
    const double* Dg = A + lt_tile(tk, tk) * kTileElems;
    tile_load_acc(Dg, lane, dacc);
    if (own_update) {
      const double* Lk = A + lt_tile(tk, j) * kTileElems;
      tile_syrk_sub(Lk, Lk, lane, dacc);
    }
    tile_store_acc(&s_T[0][0], kTile + 1, lane, dacc);
  } else if (!diag) {
    // this workgroup's own updated tile -> s_X
    v4d acc[2][2];
    tile_load_acc(C, lane, acc);
    if (own_update) tile_syrk_sub(A + lt_tile(ti, j) * kTileElems, A + lt_tile(tk, j) * kTileElems, lane, acc);
    tile_store_acc(&s_X[0][0], kTile + 1, lane, acc);
  }
  __syncthreads();
  double x[kTile];
  if (wave == 0) {
    bool ok = true;
    if (!(dbg & 1)) ok = potrf_tile(s_T, lane, s_inv, s_col, s_Lt, &s_ready);
    else if (lane == 0) __hip_atomic_store(&s_ready, kTile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    // The factored diagonal tile is NOT written back over A(tk,tk): every workgroup of this panel column
    // loads A(tk,tk) at its start, and one that is scheduled late (a busy GPU) would otherwise find L there
    // instead of the matrix.  Nothing reads L(tk,tk) from memory afterwards — the back substitution uses the
    // stored L^-T of the diagonal tiles.
    if (diag && !ok && lane == 0) atomicExch(fail, 1);
  } else {
    if (diag) {
#pragma unroll
      for (int c = 0; c < kTile; ++c) x[c] = (c == row) ? 1.0 : 0.0;  // identity: x L^-T = row of L^-T
    } else {
#pragma unroll
      for (int c = 0; c < kTile; ++c) x[c] = s_X[row][c];
    }
    if (!(dbg & 2)) trsm_row(x, s_Lt, s_inv, &s_ready);
    if (lane < kTile) {
      double2* dst = diag ? reinterpret_cast<double2*>(LinvT + (size_t)tk * kTileElems + lane * kTile)
                          : reinterpret_cast<double2*>(C + lane * kTile);
#pragma unroll
      for (int c = 0; c < kTile; c += 2) dst[c >> 1] = make_double2(x[c], x[c + 1]);
    }
  }
}

// ---- back substitution  y = L^-T z  in groups of kBsG tile rows --------------------------------
// z starts as row 0 of tile row nt (forward substitution came with the factorisation).  Groups are
// processed from the bottom; every workgroup of a launch re-solves the group (its tiles are staged
// in LDS, the diagonal solves are mat-vecs with the stored L^-T tiles), then workgroup w updates the
// slice z_w of one earlier tile column with the group's y.
constexpr int kBsG = 4;

__global__ __launch_bounds__(256) void k_z_init(const double* A, int nt, double* zbuf) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < nt * kTile) zbuf[c] = A[lt_tile(nt, c >> 5) * kTileElems + (c & 31)];
}

__global__ __launch_bounds__(256) void k_backsub_group(const double* A, const double* LinvT, int nt, int n, int t0, int t1,
                                                      double* zbuf, double* y) {
  __shared__ double s_tiles[(kBsG + kBsG * (kBsG - 1) / 2) * kTileElems];  // LinvT of the group, then L[tj][c], c<tj in group
  __shared__ double s_z[kBsG * kTile];
  __shared__ double s_y[kBsG * kTile];
  __shared__ double s_part[8 * kTile];
  const int tid = threadIdx.x;
  const int G = t1 - t0;
  // stage tiles
  for (int g = 0; g < G; ++g) {
    const double2* src = reinterpret_cast<const double2*>(LinvT + (size_t)(t0 + g) * kTileElems);
    double2* dst = reinterpret_cast<double2*>(s_tiles + g * kTileElems);
    for (int e = tid; e < kTileElems / 2; e += 256) dst[e] = src[e];
  }
  {
    int slot = kBsG;
    for (int g = 1; g < G; ++g)
      for (int c = 0; c < g; ++c, ++slot) {
        const double2* src = reinterpret_cast<const double2*>(A + lt_tile(t0 + g, t0 + c) * kTileElems);
        double2* dst = reinterpret_cast<double2*>(s_tiles + slot * kTileElems);
        for (int e = tid; e < kTileElems / 2; e += 256) dst[e] = src[e];
      }
  }
  if (tid < G * kTile) s_z[tid] = zbuf[t0 * kTile + tid];
  // prefetch this workgroup's column tiles L[t0+g][w] into registers: thread (part, q) needs rows 4 part .. +3
  const int w = blockIdx.x;
  const bool has_col = (w < t0);
  const int q = tid & 31, part = tid >> 5;
  double lc[kBsG][4];
  if (has_col) {
#pragma unroll
    for (int g = 0; g < kBsG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        lc[g][r] = (g < G) ? A[lt_tile(t0 + g, w) * kTileElems + (4 * part + r) * kTile + q] : 0.0;
  }
  __syncthreads();
  // solve the group bottom-up
  for (int g = G - 1; g >= 0; --g) {
    // y_g = LinvT_g z_g : thread (part, q): partial over columns 4 part .. +3 of row q
    {
      const double* T = s_tiles + g * kTileElems + q * kTile + 4 * part;
      const double* z = s_z + g * kTile + 4 * part;
      s_part[part * kTile + q] = T[0] * z[0] + T[1] * z[1] + T[2] * z[2] + T[3] * z[3];
    }
    __syncthreads();
    if (tid < kTile) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < 8; ++p) s += s_part[p * kTile + tid];
      s_y[g * kTile + tid] = s;
    }
    __syncthreads();
    // z_c -= L[g][c]^T y_g for c < g inside the group
    for (int idx = tid; idx < g * kTile; idx += 256) {
      const int c = idx >> 5, qq = idx & 31;
      const int slot = kBsG + g * (g - 1) / 2 + c;
      const double* Lt = s_tiles + slot * kTileElems;
      double s = 0.0;
#pragma unroll 8
      for (int r = 0; r < kTile; ++r) s += Lt[r * kTile + qq] * s_y[g * kTile + r];
      s_z[c * kTile + qq] -= s;
    }
    __syncthreads();
  }
  // update the slice of an earlier tile column
  if (has_col) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < kBsG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += (g < G) ? lc[g][r] * s_y[g * kTile + 4 * part + r] : 0.0;
    s_part[part * kTile + q] = s;
  }
  __syncthreads();
  if (has_col && tid < kTile) {
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 8; ++p) s += s_part[p * kTile + tid];
    zbuf[w * kTile + tid] -= s;
  }
  if (w == 0 && tid < G * kTile) {
    const int gidx = t0 * kTile + tid;
    if (gidx < n) y[gidx] = s_y[tid];
  }
}

// ---- y = L^-T z from the inverse accumulators (two launches) --------------------------------------------
// w_i = L(i,i)^-T z_i, z_i = row 0 of the right-hand-side tile (nt, i); y starts as w
__global__ __launch_bounds__(64) void k_inv_w(const double* A, const double* LinvT, int nt, int n, double* wbuf, double* y) {
  const int i = blockIdx.x, r = threadIdx.x & 31, h = threadIdx.x >> 5;
  const double* z = A + lt_tile(nt, i) * kTileElems;
  const double* Li = LinvT + (size_t)i * kTileElems + r * kTile;
  double sacc = 0.0;
#pragma unroll
  for (int c = 16 * h; c < 16 * h + 16; ++c) sacc += Li[c] * z[c];
  sacc += __shfl_down(sacc, 32, 64);
  if (threadIdx.x < 32) {
    wbuf[i * kTile + r] = sacc;
    if (i * kTile + r < n) y[i * kTile + r] = sacc;
  }
}
// y_k -= sum_{i>k, i = k+1+s (mod kInvSplit)} P(i,k)^T w_i : the long columns are cut into kInvSplit workgroups
constexpr int kInvSplit = 4;
__global__ __launch_bounds__(256) void k_inv_y(const double* Pinv, const double* wbuf, int nt, int n, double* y) {
  __shared__ double s_part[8][kTile];
  const int k = blockIdx.x, c = threadIdx.x & 31, part = threadIdx.x >> 5;
  if (k + 1 + (int)blockIdx.y >= nt) return;
  double sacc = 0.0;
  for (int i = k + 1 + (int)blockIdx.y; i < nt; i += kInvSplit) {
    const double* Pt = Pinv + lt_tile(i, k) * kTileElems;
    const double* w = wbuf + i * kTile;
#pragma unroll
    for (int r = 4 * part; r < 4 * part + 4; ++r) sacc += Pt[r * kTile + c] * w[r];
  }
  s_part[part][c] = sacc;
  __syncthreads();
  if (part == 0) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += s_part[q][c];
    const int g = k * kTile + c;
    if (g < n) atomicAdd(&y[g], -v);
  }
}

// ---- host wrappers -----------------------------------------------------------------------------------
void launch_assemble(const AssembleArgs& a, hipStream_t s) {
  const int64_t ntiles = (int64_t)(a.nt + 1) * (a.nt + 2) / 2;
  hipLaunchKernelGGL(k_assemble, dim3((unsigned)ntiles), dim3(256), 0, s, a);
}

// factor + forward substitution (steps -1 .. nt-2) and back substitution.
// work: nt*1024 doubles for L^-T of the diagonal tiles followed by nt*32 doubles for z.
static void launch_big(double* A, int nt, int j, int c0, int tk_lo, int tk_hi, hipStream_t s) {
  if (tk_hi < tk_lo) return;
  const int rows = nt - tk_lo + 1, cols = tk_hi - tk_lo + 1;
  hipLaunchKernelGGL(k_big_update, dim3((rows + 1) / 2, (cols + 1) / 2), dim3(256), 0, s, A, nt, j, c0, tk_lo, tk_hi);
}

// layout of the work buffer: L^-T of the diagonal tiles | z | w | inverse accumulators (plain path only)
size_t dense_work_doubles(int nt) {
  const size_t t = (size_t)(nt > 0 ? nt : 1);
  size_t n = t * kTileElems + 2 * t * kTile;
  if (nt <= kPlainMaxTiles) n += (t * (t + 1) / 2) * kTileElems;
  return n;
}
static int dense_panel_width(int nt, const DenseOverlap* ov) {
  return (ov && ov->nb > 0) ? ov->nb : ((nt <= kPlainMaxTiles) ? nt : 8);
}
// the accumulators of the inverse propagation, or NULL when this solve does not use them (outer panels, switched off)
double* dense_pinv(double* work, int nt, const DenseOverlap* ov) {
  if (nt <= 0 || nt > kPlainMaxTiles || (ov && ov->no_inverse) || dense_panel_width(nt, ov) < nt) return nullptr;
  return work + (size_t)nt * kTileElems + 2 * (size_t)nt * kTile;
}

// ov (may be NULL): a second stream and events.  With it the update of an outer panel is split: the tile
// columns of the NEXT panel are updated on the main stream (the factorisation needs them next), the columns
// beyond run on the second stream under the next panel's factorisation steps.
void launch_dense_solve(double* A, double* work, int nt, int n, double* y, int* fail, hipStream_t s, DenseOverlap* ov) {
  if (nt <= 0) return;
  double* LinvT = work;
  double* zbuf = work + (size_t)nt * kTileElems;
  double* wbuf = zbuf + (size_t)nt * kTile;
  // Outer panels of NB tile columns.  Inside a panel the plain right-looking steps run on the panel's
  // columns only; the tiles to the right then receive the whole panel in one launch (their load / store is
  // paid once per NB columns).  Up to 64 tile columns the matrix is one panel: exactly the plain algorithm.
  const int NB = dense_panel_width(nt, ov);
  const bool big_kernel = !ov || ov->big;
  const bool overlap = ov && ov->s2 && ov->overlap && big_kernel && NB < nt;
  // inverse propagation: plain path only (one panel), unless switched off for A/B measurements
  double* Pinv = dense_pinv(work, nt, ov);
  int npanel = 0;
  bool b_pending = false;  // an update on the second stream has been recorded in ov->evB and not yet waited for
  for (int p0 = 0; p0 < nt; p0 += NB, ++npanel) {
    const int pend = (p0 + NB - 1 < nt - 1) ? p0 + NB - 1 : nt - 1;
    // factor column p0 (its tiles already hold every earlier column); ti in [p0, nt]
    hipLaunchKernelGGL(k_chol_step, dim3(nt - p0 + 1, 1), dim3(128), 0, s, A, LinvT, nt, p0 - 1, fail, g_dbg_flags, p0 - 1, p0,
                       kStepNoOwnUpdate, (double*)nullptr, 1);
    for (int j = p0; j <= pend - 1; ++j) {  // apply column j to columns (j, pend], factor column j+1; ti in [j+1, nt]
      const int rows = nt - j, cols = pend - j;
      // inverse roles of launch j: (nt-1-j) rows x (j+1) columns, appended behind the tile grid
      const int extra = Pinv ? (((nt - 1 - j + kInvRows - 1) / kInvRows) * (j + 1) + rows - 1) / rows : 0;
      hipLaunchKernelGGL(k_chol_step, dim3(rows, cols + extra), dim3(128), 0, s, A, LinvT, nt, j, fail, g_dbg_flags, j, pend, 0, Pinv, cols);
    }
    if (pend >= nt - 1) break;
    // columns (pend, nt-1] receive the panel p0..pend; ti in [pend+1, nt]
    if (!big_kernel) {
      hipLaunchKernelGGL(k_chol_step, dim3(nt - pend, nt - 1 - pend), dim3(128), 0, s, A, LinvT, nt, pend, fail, g_dbg_flags, p0, nt - 1,
                         kStepBig, (double*)nullptr, nt - 1 - pend);
    } else if (!overlap) {
      launch_big(A, nt, pend, p0, pend + 1, nt - 1, s);
    } else {
      const int qend = (pend + NB < nt - 1) ? pend + NB : nt - 1;  // last column of the next panel
      hipEvent_t evF = ov->evF[npanel & 3], evB = ov->evB[npanel & 3];
      (void)hipEventRecord(evF, s);                                  // panel p0..pend is final
      // next panel's columns on the main stream; they were last written by the previous second-stream update
      if (b_pending) (void)hipStreamWaitEvent(s, ov->evB[(npanel - 1) & 3], 0);
      launch_big(A, nt, pend, p0, pend + 1, qend, s);
      // the columns beyond, concurrently with the next panel's steps (disjoint tile columns)
      if (qend < nt - 1) {
        (void)hipStreamWaitEvent(ov->s2, evF, 0);
        launch_big(A, nt, pend, p0, qend + 1, nt - 1, ov->s2);
        (void)hipEventRecord(evB, ov->s2);
        b_pending = true;
      } else {
        b_pending = false;
      }
    }
  }
  if (overlap && npanel > 0) {
    // everything queued on the second stream must be complete before the substitution (and the next assemble)
    for (int k = 0; k < 4; ++k) (void)hipStreamWaitEvent(s, ov->evB[k], 0);
  }
  if (Pinv) {
    hipLaunchKernelGGL(k_inv_w, dim3(nt), dim3(64), 0, s, A, LinvT, nt, n, wbuf, y);
    hipLaunchKernelGGL(k_inv_y, dim3(nt, kInvSplit), dim3(256), 0, s, Pinv, wbuf, nt, n, y);
    return;
  }
  hipLaunchKernelGGL(k_z_init, dim3((nt * kTile + 255) / 256), dim3(256), 0, s, A, nt, zbuf);
  for (int t1 = nt; t1 > 0; t1 -= kBsG) {
    const int t0 = t1 - kBsG > 0 ? t1 - kBsG : 0;
    hipLaunchKernelGGL(k_backsub_group, dim3(t0 > 0 ? t0 : 1), dim3(256), 0, s, A, LinvT, nt, n, t0, t1, zbuf, y);
  }
}

}  // namespace mpsfm
