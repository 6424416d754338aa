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

// In-LDS Cholesky of a 32x32 tile (row stride 33) by one wave; returns false on a non-positive pivot.
__device__ __forceinline__ bool tile_potrf_lds(double (*T)[kTile + 1], int lane) {
  bool ok = true;
  const int i = lane & 31, h = lane >> 5;
  for (int j = 0; j < kTile; ++j) {
    const double d = T[j][j];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    const double s = sqrt(d), inv = 1.0 / s;
    __syncthreads();
    if (h == 0) {
      if (i == j) T[j][j] = s;
      else if (i > j) T[i][j] *= inv;
    }
    __syncthreads();
    const double lij = T[i][j];
    for (int c = j + 1 + h; c <= i; c += 2) T[i][c] -= lij * T[c][j];
    __syncthreads();
  }
  return ok;
}

// One step of the right-looking factorisation.  j = -1: factor tile column 0 only.  Otherwise:
// every trailing tile (ti >= tk > j) gets  A[ti][tk] -= L[ti][j] L[tk][j]^T  and the tiles of column
// j+1 are turned into L right away (each of those waves re-derives and factors the diagonal tile).
__global__ __launch_bounds__(64) void k_chol_step(double* A, int nt, int j, int* fail) {
  __shared__ double s_D[kTile][kTile + 1];
  __shared__ double s_X[kTile][kTile + 1];
  const int lane = threadIdx.x;
  const int tk = j + 1 + blockIdx.y;
  const int ti = j + 1 + blockIdx.x;
  if (ti < tk || tk >= nt || ti > nt) return;
  double* C = A + lt_tile(ti, tk) * kTileElems;
  v4d acc[2][2];
  tile_load_acc(C, lane, acc);
  if (j >= 0) tile_syrk_sub(A + lt_tile(ti, j) * kTileElems, A + lt_tile(tk, j) * kTileElems, lane, acc);
  if (tk != j + 1) {
    tile_store_acc(C, kTile, lane, acc);
    return;
  }
  // panel column j+1
  if (ti == tk) {
    tile_store_acc(&s_D[0][0], kTile + 1, lane, acc);
    __syncthreads();
    const bool ok = tile_potrf_lds(s_D, lane);
    if (!ok && lane == 0) atomicExch(fail, 1);
    for (int e = lane; e < kTileElems; e += 64) {
      const int r = e >> 5, c = e & 31;
      C[e] = (c <= r) ? s_D[r][c] : 0.0;
    }
    return;
  }
  {
    v4d dacc[2][2];
    const double* Dg = A + lt_tile(tk, tk) * kTileElems;
    tile_load_acc(Dg, lane, dacc);
    if (j >= 0) {
      const double* Lk = A + lt_tile(tk, j) * kTileElems;
      tile_syrk_sub(Lk, Lk, lane, dacc);
    }
    tile_store_acc(&s_D[0][0], kTile + 1, lane, dacc);
    tile_store_acc(&s_X[0][0], kTile + 1, lane, acc);
  }
  __syncthreads();
  tile_potrf_lds(s_D, lane);
  // X <- X L^-T : row i of X solved against the lower-triangular D
  if (lane < kTile) {
    double x[kTile];
#pragma unroll
    for (int c = 0; c < kTile; ++c) x[c] = s_X[lane][c];
#pragma unroll
    for (int c = 0; c < kTile; ++c) {
      double s = x[c];
#pragma unroll
      for (int k = 0; k < c; ++k) s -= x[k] * s_D[c][k];
      x[c] = s / s_D[c][c];
    }
#pragma unroll
    for (int c = 0; c < kTile; ++c) s_X[lane][c] = x[c];
  }
  __syncthreads();
  for (int e = lane; e < kTileElems; e += 64) C[e] = s_X[e >> 5][e & 31];
}

// y = L^-T z.  z is row 0 of tile row nt.  Single workgroup; column-oriented so that the tile-row
// reads are contiguous:  for tj = nt-1 .. 0:  solve L[tj][tj]^T y_j = z_j;  z_c -= L[tj][c]^T y_j, c < tj.
__global__ __launch_bounds__(256) void k_backsub(const double* A, int nt, int n, double* y) {
  extern __shared__ double s_z[];  // nt*32 doubles + 32
  double* s_y = s_z + (size_t)nt * kTile;
  const int tid = threadIdx.x;
  for (int c = tid; c < nt * kTile; c += 256) {
    const int tj = c >> 5;
    s_z[c] = A[lt_tile(nt, tj) * kTileElems + (c & 31)];
  }
  __syncthreads();
  for (int tj = nt - 1; tj >= 0; --tj) {
    const double* D = A + lt_tile(tj, tj) * kTileElems;
    if (tid < 64) {
      // lane k keeps column k of the diagonal tile in registers; the chain runs on readlane
      // broadcasts:  y_i = z_i / D[i][i];  z_k -= D[i][k] y_i  (k < i)
      const int k = tid & 31;
      double col[kTile];
#pragma unroll
      for (int i = 0; i < kTile; ++i) col[i] = D[i * kTile + k];
      double dinv = 1.0;
#pragma unroll
      for (int i = 0; i < kTile; ++i) dinv = (i == k) ? 1.0 / col[i] : dinv;
      double zk = s_z[tj * kTile + k], yres = 0.0;
#pragma unroll
      for (int i = kTile - 1; i >= 0; --i) {
        const double t = zk * dinv;
        const double yi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), i),
                                           __builtin_amdgcn_readlane(__double2loint(t), i));
        yres = (k == i) ? yi : yres;
        zk -= col[i] * yi;
      }
      if (tid < 32) s_y[k] = yres;
    }
    __syncthreads();
    // z_c -= L[tj][c]^T y_j for all c < tj : element (col c*32+q) -= sum_r L[tj][c][r][q] * y_j[r]
    for (int idx = tid; idx < tj * kTile; idx += 256) {
      const int c = idx >> 5, q = idx & 31;
      const double* Lt = A + lt_tile(tj, c) * kTileElems;
      double s = 0.0;
#pragma unroll 8
      for (int r = 0; r < kTile; ++r) s += Lt[r * kTile + q] * s_y[r];
      s_z[idx] -= s;
    }
    if (tid < 32) {
      const int g = tj * kTile + tid;
      if (g < n) y[g] = s_y[tid];
    }
    __syncthreads();
  }
}

// ---- host wrappers -----------------------------------------------------------------------------------
void launch_assemble(const AssembleArgs& a, hipStream_t s) {
  const int64_t ntiles = (int64_t)(a.nt + 1) * (a.nt + 2) / 2;
  hipLaunchKernelGGL(k_assemble, dim3((unsigned)ntiles), dim3(256), 0, s, a);
}

// factor + forward substitution (steps -1 .. nt-2) and back substitution
void launch_dense_solve(double* A, int nt, int n, double* y, int* fail, hipStream_t s) {
  if (nt <= 0) return;
  for (int j = -1; j <= nt - 2; ++j) {
    const int rows = nt - j;      // ti in [j+1, nt]
    const int cols = (j < 0) ? 1 : nt - 1 - j;  // tk in [j+1, nt-1]; the first step only factors column 0
    hipLaunchKernelGGL(k_chol_step, dim3(rows, cols), dim3(64), 0, s, A, nt, j, fail);
  }
  const size_t lds = ((size_t)nt * kTile + kTile) * sizeof(double);
  hipLaunchKernelGGL(k_backsub, dim3(1), dim3(256), lds, s, A, nt, n, y);
}

}  // namespace mpsfm
