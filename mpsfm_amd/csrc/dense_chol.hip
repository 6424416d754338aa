// Dense solve of the reduced camera system (gfx950): what Ceres' SPARSE_SCHUR hands to CHOLMOD
// (reference call site mpsfm/sfm/mapper/bundle_adjustment.py:285-293), done here as a tiled
// right-looking Cholesky on fp64 MFMA (v_mfma_f64_16x16x4_f64).
//
// Storage: lower triangle of 32x32 tiles, tile (ti, tj) at lt_tile(ti, tj) * 1024 doubles,
// row-major inside a tile.  Tile row `nt` (one extra) carries the right-hand side in its row 0, so
// the forward substitution z = L^-1 rhs falls out of the factorisation; k_backsub then solves
// L^T y = z.
#include "common.h"
#include "dense_tile.h"

namespace mpsfm {

constexpr int kPlainMaxTiles = 64;  // up to this many tile columns: one outer panel, inverse propagation instead of back substitution

// four elements (rows r0 + k * rstep, k < 4, column c) of tile (ti, tj) of the damped reduced system [S + D / radius | rhs row];
// tile row nt carries the right-hand side in its row 0.  Branch-free per element, so that the three dependent loads
// (column -> slot, slot pair -> block, block -> value) of the four elements overlap.
__device__ __forceinline__ void assemble_quad(const AssembleArgs& P, double lm_radius, int ti, int tj, int r0, int rstep, int c, double out[4]) {
  const int n = P.n;
  const int C = tj * kTile + c;
  const int ic = vec_index(P.col_slot, C, n);
  if (ti == P.nt) {
    const double rhs = (tj < P.nt && ic >= 0) ? P.wv[ic] - P.gc[ic] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (r0 + k * rstep == 0) ? rhs : 0.0;
    return;
  }
  const int icv = ic >= 0 ? ic : 0;
  const int bc = icv / 6, b = icv - bc * 6;
  int ir[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ir[k] = vec_index(P.col_slot, ti * kTile + r0 + k * rstep, n);
  int64_t off[4];
  bool has[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool real = ir[k] >= 0 && ic >= 0;
    const int irv = ir[k] >= 0 ? ir[k] : 0;
    const int br = irv / 6, a = irv - br * 6;
    const int lo = br < bc ? br : bc, hi = br < bc ? bc : br;
    int64_t blk;
    if (P.sky.index) blk = P.sky.index[(int64_t)hi * P.sky.ns + lo];
    else blk = lo >= P.sky.first[hi] ? P.sky.start[hi] + (lo - P.sky.first[hi]) : -1;
    has[k] = real && blk >= 0;  // no landmark shared by the two cameras: structurally zero
    // the stored block is (lo, hi): transposed when the row's camera comes second; the diagonal block holds its upper triangle
    const int e = br < bc ? a * 6 + b : br > bc ? b * 6 + a : (a <= b ? a * 6 + b : b * 6 + a);
    off[k] = has[k] ? blk * 36 + e : 0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double v = P.Sblk[off[k]];
    const int R = ti * kTile + r0 + k * rstep;
    const bool real = ir[k] >= 0 && ic >= 0;
    out[k] = has[k] ? v : (!real && R == C) ? 1.0 : 0.0;  // padding columns (tile alignment of the dissection's parts, tail of the last tile): identity
    if (real && R == C) out[k] += fmin(fmax(P.diagU[ir[k]], P.min_diag), P.max_diag) / lm_radius;
  }
}
// one workgroup (256 threads): tile (ti, tj) of the system and / or a zero for its inverse accumulator
__device__ __forceinline__ void assemble_tile(const AssembleArgs& P, double lm_radius, int ti, int tj, bool write) {
  const int64_t id = lt_tile(ti, tj);
  if (P.Pinv && ti < P.nt && tj < ti)
    for (int e = threadIdx.x; e < kTileElems; e += 256) P.Pinv[id * kTileElems + e] = 0.0;
  if (!write) return;
  double* T = P.A + id * kTileElems;
  const int e = threadIdx.x;  // elements e + 256 k: rows (e >> 5) + 8 k, column e & 31
  double v[4];
  assemble_quad(P, lm_radius, ti, tj, e >> 5, 8, e & 31, v);
#pragma unroll
  for (int k = 0; k < 4; ++k) T[e + 256 * k] = v[k];
}

// one workgroup per lower tile (including the rhs tile row)
__global__ __launch_bounds__(256) void k_assemble(AssembleArgs P) {
  if (lm_over(P.ctl)) return;
  const double lm_radius = P.ctl ? lm_radius_of(P.ctl) : P.radius;
  // decode tile id -> (ti, tj), ti >= tj, ti in [0, nt], tj in [0, nt-1]
  const int64_t id = P.tile_list ? (int64_t)P.tile_list[blockIdx.x] : (int64_t)blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
  while ((int64_t)ti * (ti + 1) / 2 > id) --ti;
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= id) ++ti;
  const int tj = (int)(id - (int64_t)ti * (ti + 1) / 2);
  if (P.y && blockIdx.x == 0)  // the solution vector is accumulated with atomics (k_inv_y): start from zero
    for (int e = threadIdx.x; e < P.n_vec; e += 256) P.y[e] = 0.0;
  assemble_tile(P, lm_radius, ti, tj, !(P.live && !P.live[id]));
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
// Phase timeline of the factorisation (diagnostics, scripts/dbg_chol_trace.py): when a buffer is registered, the
// workgroup that owns the diagonal tile of every step stores wall_clock64() (100 MHz) at its phase boundaries,
// 8 stamps per wave and step.
__device__ long long* g_chol_trace = nullptr;
extern "C" int mpsfm_debug_set_chol_trace(long long* dev_buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_trace), &dev_buf, sizeof(dev_buf));
}
#define MPSFM_STAMP(k) do { if (trace_on && lane == 0) tr[(k)] = wall_clock64(); } while (0)

// Dense outer-panel path (reduced systems without exploitable structure, MPSFM_CHOL_NB): one step of the right-looking
// factorisation, four waves per tile.  j = -1: factor tile column 0 only.
//   trailing tile (tk > j+1):  A[ti][tk] -= L[ti][j] L[tk][j]^T, one quadrant per wave.
//   panel tile (tk == j+1):    every workgroup re-derives the updated diagonal tile; wave 0 factors the stacked
//                              [D; X] (stacked_panel); the workgroup that owns the diagonal tile stores L^-T
//                              (kept for the back substitution).
// c0 is the first tile column whose L is applied to a trailing tile in this
// launch (c0 == j: the plain right-looking step), tk_max the last tile column this launch touches, and
//   kStepNoOwnUpdate  the panel column j+1 has already received column j (first step of an outer panel),
//   kStepBig          no factorisation: every tile (ti, tk), j < tk <= tk_max, gets columns c0..j at once
//                     (one load and one store of the tile for a rank-32*(j-c0+1) update).
constexpr int kStepNoOwnUpdate = 1, kStepBig = 2;
// Inverse propagation (level-scheduled path up to kPlainMaxTiles tile columns, Pinv != NULL).  The back substitution
// y = L^-T z is another chain of dependent tile solves, one launch per level.  Instead the launches also build, in the
// shadow of their latency-bound panel factorisation, the accumulators  P(i,k) = sum_{j=k}^{i-1} L(i,j) X(j,k)  of the
// inverse X = L^-1  (X(k,k) = L(k,k)^-1, X(i,k) = -L(i,i)^-1 P(i,k)), after which
//   y_k = w_k - sum_{i>k} P(i,k)^T w_i,   w_i = L(i,i)^-T z_i
// is two launches.  The launch after column j was factored uses column j of L and L(j,j)^-1, both final:
//   role (j, k), k in the subtree of j (X(j,k) is zero elsewhere):   P(i,k) += L(i,j) X(j,k) for the rows i of struct(j),
//   X(j,k) recomputed from P(j,k) (final: its contributions come from descendants of j, whose roles ran in earlier
//   launches), so no workgroup reads what another one writes in the same launch; two columns of one level have disjoint
//   subtrees, so their roles never write the same accumulator.
#ifndef MPSFM_INV_ROWS
#define MPSFM_INV_ROWS 2
#endif
constexpr int kInvRows = MPSFM_INV_ROWS;  // rows i handled by one inverse-role workgroup (X(j,k) is formed once for all of them)
constexpr int kStepThreads = 256;         // four waves: one 16 x 16 quadrant of a tile each
__device__ __forceinline__ void inv_role(const double* A, const double* LinvT, double* Pinv, int nt, int j, const int* irow, int ni_rows, int k,
                                         double (*s_B)[kTile + 1], double* s_C) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = wave & 1, ni = wave >> 1;
  const int m16 = lane & 15, kg = lane >> 4;
  const double* Li = LinvT + (size_t)j * kTileElems;               // L(j,j)^-T, row-major
  const double* Pjk = Pinv + lt_tile(j, k) * kTileElems;           // only read when k < j
  // accumulators and A operands of every row of this workgroup are requested now: their latency runs under the
  // formation of T.  Lane (m, q) reads L(i,j)[16 mi + m][4 s + kg], s = 0..7.
  v4d acc[kInvRows];
  double aop[kInvRows][8];
#pragma unroll
  for (int r = 0; r < kInvRows; ++r) {
    const int i = irow[min(r, ni_rows - 1)];
    quad_load(Pinv + lt_tile(i, k) * kTileElems, kTile, lane, mi, ni, acc[r]);
    const double* arow = A + lt_tile(i, j) * kTileElems + (16 * mi + m16) * kTile + kg;
#pragma unroll
    for (int s = 0; s < 8; ++s) aop[r][s] = arow[4 * s];
  }
  for (int e = tid; e < kTileElems; e += kStepThreads) {
    const int r = e >> 5, c = e & 31;
    s_B[r][c] = Li[e];
    s_C[e] = (k < j) ? Pjk[e] : 0.0;
  }
  __syncthreads();
  // T = X(j,k): k < j: -(L(j,j)^-1 P(j,k)),  T[m][n] = -sum_q LinvT[q][m] Pjk[q][n];  k == j: L(j,j)^-1, T[m][n] = LinvT[n][m]
  v4d t = {0, 0, 0, 0};
  if (k < j) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
      t = __builtin_amdgcn_mfma_f64_16x16x4f64(-s_B[4 * s + kg][16 * mi + m16], s_C[(4 * s + kg) * kTile + 16 * ni + m16], t, 0, 0, 0);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = s_B[16 * ni + m16][16 * mi + kg + 4 * r];
  }
  __syncthreads();  // T overwrites L(j,j)^-T in LDS once every wave has consumed it
  quad_store(&s_B[0][0], kTile + 1, lane, mi, ni, t);
  __syncthreads();
  // P(i,k) += L(i,j) T for the rows of this workgroup; the B operand (T) stays in registers
  double bT[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bT[s] = s_B[4 * s + kg][16 * ni + m16];
#pragma unroll
  for (int r = 0; r < kInvRows; ++r) {
    if (r >= ni_rows) break;
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[r][s], bT[s], acc[r], 0, 0, 0);
    quad_store(Pinv + lt_tile(irow[r], k) * kTileElems, kTile, lane, mi, ni, acc[r]);
  }
}

__global__ __launch_bounds__(kStepThreads) void k_chol_step(double* A, double* LinvT, int nt, int j, int* fail, int dbg, int c0, int tk_max,
                                                            int mode) {
  __shared__ double s_T[kTile][kTile + 1];
  __shared__ double s_X[kTile][kTile + 1];
  __shared__ __attribute__((aligned(16))) double s_Lt[kTile * kTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mi = wave & 1, ni = wave >> 1;  // this wave's quadrant
  if (dbg & 16) return;  // ablation: the launch chain alone
  const int tk = j + 1 + blockIdx.y, ti = j + 1 + blockIdx.x;
  if (ti < tk || tk >= nt || ti > nt || tk > tk_max) return;
  double* C = A + lt_tile(ti, tk) * kTileElems;
  if (tk != j + 1 || (mode & kStepBig)) {
    // trailing tile: A[ti][tk] -= sum_c L[ti][c] L[tk][c]^T, one quadrant per wave
    v4d acc;
    quad_load(C, kTile, lane, mi, ni, acc);
    if (!(dbg & 4))
      for (int c = c0; c <= j; ++c) {
        double a[8], b[8];
        quad_operand(A + lt_tile(ti, c) * kTileElems, lane, mi, a);
        quad_operand(A + lt_tile(tk, c) * kTileElems, lane, ni, b);
        quad_gemm_sub(a, b, acc);
      }
    quad_store(C, kTile, lane, mi, ni, acc);
    return;
  }
  const bool own_update = (j >= 0) && !(mode & kStepNoOwnUpdate) && !(dbg & 4);
  // ---- panel column j+1 -------------------------------------------------------------------
  const int row = lane & 31;
  const bool diag = (ti == tk);
  long long* tr = g_chol_trace ? g_chol_trace + ((size_t)(j + 1) * 2) * 8 : nullptr;
  const bool trace_on = diag && tr != nullptr && wave == 0;
  MPSFM_STAMP(0);
  {
    // every wave: its quadrant of the updated diagonal tile -> s_T and of the workgroup's own updated tile -> s_X
    v4d dacc, xacc;
    quad_load(A + lt_tile(tk, tk) * kTileElems, kTile, lane, mi, ni, dacc);
    if (!diag) quad_load(C, kTile, lane, mi, ni, xacc);
    if (own_update) {
      const double* Lk = A + lt_tile(tk, j) * kTileElems;
      double am[8], bn[8];
      quad_operand(Lk, lane, mi, am);
      quad_operand(Lk, lane, ni, bn);
      quad_gemm_sub(am, bn, dacc);
      if (!diag) {
        quad_operand(A + lt_tile(ti, j) * kTileElems, lane, mi, am);
        quad_gemm_sub(am, bn, xacc);
      }
    }
    quad_store(&s_T[0][0], kTile + 1, lane, mi, ni, dacc);
    if (diag) {  // identity: x L^-T = row of L^-T
#pragma unroll
      for (int r = 0; r < 4; ++r) xacc[r] = (16 * mi + (lane >> 4) + 4 * r == 16 * ni + (lane & 15)) ? 1.0 : 0.0;
    }
    quad_store(&s_X[0][0], kTile + 1, lane, mi, ni, xacc);
  }
  MPSFM_STAMP(1);
  __syncthreads();
  MPSFM_STAMP(2);
  if (wave != 0) return;
  // stacked factorisation by wave 0: lanes 0..31 rows of D, lanes 32..63 rows of X (identity for the diagonal workgroup)
  double a[kTile];
  {
    const double* src = (lane < kTile) ? &s_T[row][0] : &s_X[row][0];
#pragma unroll
    for (int c = 0; c < kTile; ++c) a[c] = src[c];
  }
  bool ok = true;
  if (!(dbg & 1)) stacked_panel<0>(a, lane, s_Lt, ok);
  MPSFM_STAMP(3);
  // The factored diagonal tile is NOT written back over A(tk,tk): every workgroup of this panel column loads
  // A(tk,tk) at its start, and one that is scheduled late (a busy GPU) would otherwise find L there instead of the
  // matrix.  Nothing reads L(tk,tk) from memory afterwards — the back substitution uses the stored L^-T of the
  // diagonal tiles.
  if (diag && !ok && lane == 0) atomicExch(fail, 1);
  if (lane >= kTile) {
    double2* dst = diag ? reinterpret_cast<double2*>(LinvT + (size_t)tk * kTileElems + row * kTile)
                        : reinterpret_cast<double2*>(C + row * kTile);
#pragma unroll
    for (int c = 0; c < kTile; c += 2) dst[c >> 1] = make_double2(a[c], a[c + 1]);
  }
  MPSFM_STAMP(4);
}


// ---- level-scheduled factorisation (chol_plan.h): one launch per level of the tile elimination tree --------------------
// The same three kinds of workgroup as a right-looking step (k_chol_step with kStepEnv), driven by an item table:
//   panel (ti, c)   the tile's column c is at this launch's level.  The updated diagonal tile D(c) and the own tile receive
//                   the children of c that were factored in the previous launch (sources; the rest of the subtree came
//                   through trailing items of earlier launches), then wave 0 factors the stacked [D(c); X];
//   trail (ti, tk)  C -= sum over the source columns (all of the previous level that reach the tile: one workgroup per
//                   tile, so independent chains never race on an ancestor's tile and the sum has a fixed order);
//   role  (j, k)    inverse propagation for column j of the previous level and a column k of its subtree.
struct LevelArgs {
  double* A; double* LinvT; double* Pinv;
  const CholItem* items; const int32_t* srcs; const int32_t* rows;
  int* fail; int32_t nt, dbg;
  const LmCtl* ctl;
};

__global__ __launch_bounds__(kStepThreads) void k_chol_level(LevelArgs G) {
  if (lm_over(G.ctl)) return;
  __shared__ double s_T[kTile][kTile + 1];
  __shared__ double s_X[kTile][kTile + 1];
  __shared__ __attribute__((aligned(16))) double s_Lt[kTile * kTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mi = wave & 1, ni = wave >> 1;  // this wave's quadrant
  if (G.dbg & 16) return;  // ablation: the launch chain alone
  const CholItem it = G.items[blockIdx.x];
  double* A = G.A;
  if (it.type == kItemRole) {
    if (G.dbg & 8) return;
    inv_role(A, G.LinvT, G.Pinv, G.nt, it.ti, G.rows + it.aux, it.nsrc, it.tk, s_T, s_Lt);
    return;
  }
  const int ti = it.ti, tk = it.tk, nsrc = it.nsrc;
  const int32_t* src = G.srcs + it.src;
  double* C = A + lt_tile(ti, tk) * kTileElems;
  if (it.type == kItemTrail) {
    v4d acc;
    quad_load(C, kTile, lane, mi, ni, acc);
    if (!(G.dbg & 4))
      for (int q = 0; q < nsrc; ++q) {
        const int c = src[q];
        double a[8], b[8];
        quad_operand(A + lt_tile(ti, c) * kTileElems, lane, mi, a);
        quad_operand(A + lt_tile(tk, c) * kTileElems, lane, ni, b);
        quad_gemm_sub(a, b, acc);
      }
    quad_store(C, kTile, lane, mi, ni, acc);
    return;
  }
  // ---- panel tile of column tk ---------------------------------------------------------------
  const int row = lane & 31;
  const bool diag = (ti == tk);
  long long* tr = g_chol_trace ? g_chol_trace + ((size_t)tk * 2) * 8 : nullptr;
  const bool trace_on = diag && tr != nullptr && wave == 0;
  MPSFM_STAMP(0);
  {
    v4d dacc, xacc;
    quad_load(A + lt_tile(tk, tk) * kTileElems, kTile, lane, mi, ni, dacc);
    if (!diag) quad_load(C, kTile, lane, mi, ni, xacc);
    if (!(G.dbg & 4))
      for (int q = 0; q < nsrc; ++q) {
        const int e = src[q], c = e & 0xffff;
        const double* Lk = A + lt_tile(tk, c) * kTileElems;
        double am[8], bn[8];
        quad_operand(Lk, lane, mi, am);
        quad_operand(Lk, lane, ni, bn);
        quad_gemm_sub(am, bn, dacc);
        if (!diag && (e & kSrcX)) {
          quad_operand(A + lt_tile(ti, c) * kTileElems, lane, mi, am);
          quad_gemm_sub(am, bn, xacc);
        }
      }
    quad_store(&s_T[0][0], kTile + 1, lane, mi, ni, dacc);
    if (diag) {  // identity: x L^-T = row of L^-T
#pragma unroll
      for (int r = 0; r < 4; ++r) xacc[r] = (16 * mi + (lane >> 4) + 4 * r == 16 * ni + (lane & 15)) ? 1.0 : 0.0;
    }
    quad_store(&s_X[0][0], kTile + 1, lane, mi, ni, xacc);
  }
  MPSFM_STAMP(1);
  __syncthreads();
  MPSFM_STAMP(2);
  if (wave != 0) return;
  double a[kTile];
  {
    const double* sp = (lane < kTile) ? &s_T[row][0] : &s_X[row][0];
#pragma unroll
    for (int c = 0; c < kTile; ++c) a[c] = sp[c];
  }
  bool ok = true;
  if (!(G.dbg & 1)) stacked_panel<0>(a, lane, s_Lt, ok);
  MPSFM_STAMP(3);
  // as in k_chol_step the factored diagonal tile is not written back over A(tk,tk): its column's other workgroups may
  // still be loading it
  if (diag && !ok && lane == 0) atomicExch(G.fail, 1);
  if (lane >= kTile) {
    double2* dst = diag ? reinterpret_cast<double2*>(G.LinvT + (size_t)tk * kTileElems + row * kTile)
                        : reinterpret_cast<double2*>(C + row * kTile);
#pragma unroll
    for (int c = 0; c < kTile; c += 2) dst[c >> 1] = make_double2(a[c], a[c + 1]);
  }
  MPSFM_STAMP(4);
}

// Backward substitution y = L^-T z by levels, highest first (no inverse accumulators: nt > kPlainMaxTiles).  One workgroup
// per tile column j of the level: v = z_j - sum_{i in struct(j)} L(i,j)^T y_i (all those i are ancestors: final), then
// y_j = L(j,j)^-T v with the stored inverse.  z_j is row 0 of the right-hand-side tile (nt, j).
__global__ __launch_bounds__(256) void k_back_level(const double* A, const double* LinvT, const int32_t* cols, const int32_t* struct_start,
                                                    const int32_t* struct_rows, int nt, int n, double* ybuf, double* y, const int32_t* col_slot, const LmCtl* ctl) {
  __shared__ double s_part[8][kTile];
  __shared__ double s_v[kTile];
  if (lm_over(ctl)) return;
  const int j = cols[blockIdx.x];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int r0 = struct_start[j], r1 = struct_start[j + 1] - 1;  // the last entry is the right-hand-side row
  double sacc = 0.0;
  for (int q = r0 + part; q < r1; q += 8) {
    const int i = struct_rows[q];
    const double* Tl = A + lt_tile(i, j) * kTileElems + c;
    const double* yi = ybuf + (size_t)i * kTile;
#pragma unroll 8
    for (int r = 0; r < kTile; ++r) sacc = __builtin_fma(Tl[r * kTile], yi[r], sacc);
  }
  s_part[part][c] = sacc;
  __syncthreads();
  if (part == 0) {
    double v = A[lt_tile(nt, j) * kTileElems + c];
#pragma unroll
    for (int q = 0; q < 8; ++q) v -= s_part[q][c];
    s_v[c] = v;
  }
  __syncthreads();
  if (part == 0) {
    const double* Li = LinvT + (size_t)j * kTileElems + c * kTile;
    double yr = 0.0;
#pragma unroll
    for (int q = 0; q < kTile; ++q) yr = __builtin_fma(Li[q], s_v[q], yr);
    ybuf[(size_t)j * kTile + c] = yr;
    const int iy = vec_index(col_slot, j * kTile + c, n);
    if (iy >= 0) y[iy] = yr;
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
                                                      double* zbuf, double* y, const int32_t* col_slot) {
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
    const int iy = vec_index(col_slot, gidx, n);
    if (iy >= 0) y[iy] = s_y[tid];
  }
}

// ---- y = L^-T z from the inverse accumulators (one launch) ----------------------------------------------
// y_k = w_k - sum_{i>k} P(i,k)^T w_i,  w_i = L(i,i)^-T z_i,  z_i = row 0 of the right-hand-side tile (nt, i).  The long columns
// are cut into kInvSplit workgroups (i = k+1+s mod kInvSplit); every workgroup first forms the w_i it needs itself (a
// 32 x 32 product each — cheaper than a launch of its own for them), split 0 also w_k.  y starts from zero (k_assemble).
constexpr int kInvSplit = 4;
__global__ __launch_bounds__(256) void k_inv_y(const double* A, const double* LinvT, const double* Pinv, int nt, int n, double* y,
                                               const int32_t* col_slot, const LmCtl* ctl) {
  __shared__ double s_w[(kPlainMaxTiles / kInvSplit + 2) * kTile];
  __shared__ double s_part[8][kTile];
  if (lm_over(ctl)) return;
  const int k = blockIdx.x, sp = blockIdx.y, c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int first = k + 1 + sp;
  const int m = first < nt ? (nt - first + kInvSplit - 1) / kInvSplit : 0;  // tiles i = first + kInvSplit j
  const int own = sp == 0 ? 1 : 0;                                          // slot 0: w_k
  if (m + own == 0) return;
  for (int idx = threadIdx.x; idx < (m + own) * kTile; idx += 256) {
    const int slot = idx >> 5, r = idx & 31;
    const int i = (own && slot == 0) ? k : first + kInvSplit * (slot - own);
    const double* z = A + lt_tile(nt, i) * kTileElems;
    const double* Li = LinvT + (size_t)i * kTileElems + r * kTile;
    double sacc = 0.0;
#pragma unroll 8
    for (int q = 0; q < kTile; ++q) sacc = __builtin_fma(Li[q], z[q], sacc);
    s_w[idx] = sacc;
  }
  __syncthreads();
  double sacc = 0.0;
  for (int j = 0; j < m; ++j) {
    const double* Pt = Pinv + lt_tile(first + kInvSplit * j, k) * kTileElems;
    const double* w = s_w + (j + own) * kTile;
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
    const int iy = vec_index(col_slot, g, n);
    if (iy >= 0) atomicAdd(&y[iy], (own ? s_w[c] : 0.0) - v);
  }
}

// ---- host wrappers -----------------------------------------------------------------------------------
void launch_assemble(const AssembleArgs& a, hipStream_t s) {
  const int64_t ntiles = a.tile_list ? (int64_t)a.n_list : (int64_t)(a.nt + 1) * (a.nt + 2) / 2;
  hipLaunchKernelGGL(k_assemble, dim3((unsigned)ntiles), dim3(256), 0, s, a);
}

// factor + forward substitution (steps -1 .. nt-2) and back substitution.
// work: nt*1024 doubles for L^-T of the diagonal tiles followed by nt*32 doubles for z.
static void launch_big(double* A, int nt, int j, int c0, int tk_lo, int tk_hi, hipStream_t s) {
  if (tk_hi < tk_lo) return;
  const int rows = nt - tk_lo + 1, cols = tk_hi - tk_lo + 1;
  hipLaunchKernelGGL(k_big_update, dim3((rows + 1) / 2, (cols + 1) / 2), dim3(256), 0, s, A, nt, j, c0, tk_lo, tk_hi);
}

// layout of the work buffer: L^-T of the diagonal tiles | z | w | one spare tile row | inverse accumulators (per-step path, nt <= 64)
size_t dense_work_doubles(int nt) {
  const size_t t = (size_t)(nt > 0 ? nt : 1);
  size_t n = t * kTileElems + 2 * t * kTile + (t + 1) * kTileElems;
  if (nt <= kPlainMaxTiles) n += (t * (t + 1) / 2) * kTileElems;
  return n;
}
int dense_plain_max_tiles() { return kPlainMaxTiles; }
int dense_inv_rows() { return kInvRows; }
bool dense_level(const DenseOverlap* ov, const LevelPlanDev* lp) { return lp && lp->valid && !(ov && (ov->nb > 0 || ov->no_level)); }
static int dense_panel_width(int nt, const DenseOverlap* ov) {
  return (ov && ov->nb > 0) ? ov->nb : ((nt <= kPlainMaxTiles) ? nt : 8);
}
// the accumulators of the inverse propagation, or NULL when this solve does not use them (outer panels, switched off)
double* dense_pinv(double* work, int nt, const DenseOverlap* ov, const LevelPlanDev* lp) {
  if (nt <= 0 || nt > kPlainMaxTiles || (ov && (ov->no_inverse || ov->nb > 0))) return nullptr;
  if (dense_level(ov, lp) && !lp->use_pinv) return nullptr;
  return work + (size_t)nt * kTileElems + 2 * (size_t)nt * kTile + ((size_t)nt + 1) * kTileElems;
}

// ov (may be NULL): a second stream and events.  With it the update of an outer panel is split: the tile
// columns of the NEXT panel are updated on the main stream (the factorisation needs them next), the columns
// beyond run on the second stream under the next panel's factorisation steps.
void launch_dense_solve(double* A, double* work, int nt, int n, double* y, int* fail, hipStream_t s, DenseOverlap* ov, const LevelPlanDev* lp,
                        const LmCtl* ctl) {
  if (nt <= 0) return;
  double* LinvT = work;
  double* zbuf = work + (size_t)nt * kTileElems;
  if (dense_level(ov, lp)) {
    double* Pinv = dense_pinv(work, nt, ov, lp);
    LevelArgs G{A, LinvT, Pinv, nullptr, lp->d_srcs, lp->d_rows, fail, nt, g_dbg_flags, ctl};
    for (int l = 0; l < lp->nlevels; ++l) {
      const int grid = lp->h_launch_start[l + 1] - lp->h_launch_start[l];
      if (grid <= 0) continue;
      G.items = lp->d_items + lp->h_launch_start[l];
      hipLaunchKernelGGL(k_chol_level, dim3((unsigned)grid), dim3(kStepThreads), 0, s, G);
    }
    if (Pinv) {
      hipLaunchKernelGGL(k_inv_y, dim3(nt, kInvSplit), dim3(256), 0, s, A, LinvT, Pinv, nt, n, y, lp->d_col_slot, ctl);
      return;
    }
    for (int l = 0; l < lp->nlevels; ++l) {
      const int grid = lp->h_back_start[l + 1] - lp->h_back_start[l];
      if (grid <= 0) continue;
      hipLaunchKernelGGL(k_back_level, dim3((unsigned)grid), dim3(256), 0, s, A, LinvT, lp->d_back_cols + lp->h_back_start[l], lp->d_struct_start,
                         lp->d_struct_rows, nt, n, zbuf, y, lp->d_col_slot, ctl);
    }
    return;
  }
  // Outer panels of NB tile columns.  Inside a panel the plain right-looking steps run on the panel's
  // columns only; the tiles to the right then receive the whole panel in one launch (their load / store is
  // paid once per NB columns).  Up to 64 tile columns the matrix is one panel: exactly the plain algorithm.
  const int NB = dense_panel_width(nt, ov);
  const bool big_kernel = !ov || ov->big;
  const bool overlap = ov && ov->s2 && ov->overlap && big_kernel && NB < nt;
  int npanel = 0;
  bool b_pending = false;  // an update on the second stream has been recorded in ov->evB and not yet waited for
  for (int p0 = 0; p0 < nt; p0 += NB, ++npanel) {
    const int pend = (p0 + NB - 1 < nt - 1) ? p0 + NB - 1 : nt - 1;
    // factor column p0 (its tiles already hold every earlier column); ti in [p0, nt]
    hipLaunchKernelGGL(k_chol_step, dim3(nt - p0 + 1, 1), dim3(kStepThreads), 0, s, A, LinvT, nt, p0 - 1, fail, g_dbg_flags, p0 - 1, p0,
                       kStepNoOwnUpdate);
    for (int j = p0; j <= pend - 1; ++j) {  // apply column j to columns (j, pend], factor column j+1; ti in [j+1, nt]
      const int rows = nt - j, cols = pend - j;
      hipLaunchKernelGGL(k_chol_step, dim3(rows, cols), dim3(kStepThreads), 0, s, A, LinvT, nt, j, fail, g_dbg_flags, j, pend, 0);
    }
    if (pend >= nt - 1) break;
    // columns (pend, nt-1] receive the panel p0..pend; ti in [pend+1, nt]
    if (!big_kernel) {
      hipLaunchKernelGGL(k_chol_step, dim3(nt - pend, nt - 1 - pend), dim3(kStepThreads), 0, s, A, LinvT, nt, pend, fail, g_dbg_flags, p0, nt - 1,
                         kStepBig);
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
  hipLaunchKernelGGL(k_z_init, dim3((nt * kTile + 255) / 256), dim3(256), 0, s, A, nt, zbuf);
  for (int t1 = nt; t1 > 0; t1 -= kBsG) {
    const int t0 = t1 - kBsG > 0 ? t1 - kBsG : 0;
    hipLaunchKernelGGL(k_backsub_group, dim3(t0 > 0 ? t0 : 1), dim3(256), 0, s, A, LinvT, nt, n, t0, t1, zbuf, y, lp ? lp->d_col_slot : nullptr);
  }
}

}  // namespace mpsfm
