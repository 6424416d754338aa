// Single-launch Levenberg-Marquardt for SMALL problems (gfx950): the local bundle adjustments the reference runs after every
// registered image (mpsfm/sfm/mapper/bundle_adjustment.py:188-293 through pyceres.solve, a handful of cameras and a few thousand
// landmarks).  The launch chain of ba_solver.hip spends such a problem's iteration on ~12 launch boundaries; here ONE cooperative
// launch runs the whole trust-region loop, one workgroup per landmark chunk, two grid barriers per iteration:
//
//   A  track sweep of the workgroup's chunk (dense_sweep_chunk, the body of k_track_sweep_dense) at the current state; the chunk's
//      slab is added into a dense accumulator of the reduced system (<= 96 x 96) with device-scope atomics
//   -- grid barrier 1 --
//   B  EVERY workgroup loads the reduced system, damps it and solves it in LDS (right-looking tile Cholesky, the stacked one-wave
//      panel factorisation of dense_tile.h, forward substitution riding along as one more stacked row, block back substitution
//      with the stored L(j,j)^-T) — redundant, but it saves the barrier a single solving workgroup would need
//   C  candidate cameras in LDS (each workgroup keeps the variable cameras' state and table rows itself)
//   D  update sweep of the chunk (update_sweep_chunk, the body of k_update_sweep): candidate landmarks, model cost change,
//      candidate cost; the other accumulator is zeroed for the next iteration
//   -- grid barrier 2 --
//   E  every workgroup sums the chunks' partial rows in the same order and takes the same decision (lm_decide.h); an accepted
//      candidate becomes the state (cameras in LDS, the chunk's own landmarks in HBM)
//
// Every spin is bounded: a barrier that does not complete raises the abort flag, every workgroup leaves, the host reports an error.
#include "local_lm.h"
#include "sweep_common.h"
#include "sweep_dense_body.h"
#include "sweep_update_body.h"
#include "dense_tile.h"
#include "lm_decide.h"

namespace mpsfm {

namespace {
constexpr int kLT = kLocalN / kTile;                 // 3 tile columns
constexpr int kTLd = 34;                             // LDS row stride of a tile: rows 16-byte aligned, banks spread
constexpr int kTileLds = kTile * kTLd;
constexpr int kSpinMax = 1 << 21;
static_assert(kLocalN % kTile == 0 && kLT == 3, "the tile decode below assumes three tile columns");

struct DenseSolveLds {
  double T[kLT * (kLT + 1) / 2][kTileLds];   // lower tile triangle of the damped reduced system, then of L
  double Linv[kLT][kTileLds];                // L(j,j)^-T
  double P[4][kSP * kTile];                  // stacked_panel's column store, one per wave
};
union PhaseLds {
  DenseLds sweep;
  UpdLds upd;
  DenseSolveLds dense;
};
constexpr int kPer3Decl = (kLocalCams * (kLocalCams + 1) / 2 * 36 + kThreads - 1) / kThreads;
struct LocalState {
  LmHead L;
  double tab[kLocalCams * kCamRec], tab2[kLocalCams * kCamRec];   // camera table rows by slot: state, candidate
  double q[kLocalCams * 4], t[kLocalCams * 3], q2[kLocalCams * 4], t2[kLocalCams * 3];
  double cs[kLocalCams * 6];
  double y[kLocalN], z[kLocalN], rhs[kLocalN], gcv[kLocalN], v2[kLocalN], v[kTile], vp[8 * kTile];
  double camred[kLocalCams * 3];
  double red[kThreads / 64], out[8];
  int32_t cam_of_slot[kLocalCams];
  int16_t rc[kPer3Decl * kThreads];   // gather_system's element list
  int32_t flag, fail, epoch, pad_;
};

// Grid barrier over the co-resident workgroups: a monotonic arrival counter.  Every wave first waits for its own stores and
// atomics to be acknowledged (a workgroup barrier alone only waits for LDS), then ONE thread per workgroup releases at device
// scope (L2 write-back), arrives, polls, and acquires (cache invalidate) for the CU — a fence per wave costs an L2 walk each.
__device__ __forceinline__ bool grid_barrier(int32_t* bar, LocalState& S, int nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(&bar[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    S.epoch += 1;
    const int target = S.epoch * nwg;
    bool ok = false;
    for (int spin = 0; spin < kSpinMax; ++spin) {
      if (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
      if ((spin & 255) == 255 && __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store(&bar[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    S.flag = ok ? 1 : 0;
  }
  __syncthreads();
  return S.flag != 0;
}

__device__ __forceinline__ double* tile_at(DenseSolveLds& D, int i, int j) { return D.T[i * (i + 1) / 2 + j]; }
__device__ __forceinline__ void quad_operand_ld(const double* T, int lane, int b, double (&o)[8]) {
  const double2* p = reinterpret_cast<const double2*>(T + (16 * b + (lane & 15)) * kTLd + 8 * (lane >> 4));
#pragma unroll
  for (int s = 0; s < 4; ++s) { const double2 x = p[s]; o[2 * s] = x.x; o[2 * s + 1] = x.y; }
}

// elements of the upper block triangle a thread moves from the accumulator to the tiles, for one / two / three tile columns
// (at most 5 / 10 / 16 variable cameras); S.rc lists them (row << 8 | column, -1: none), filled once per launch
constexpr int per_thread(int cams) { return (cams * (cams + 1) / 2 * 36 + kThreads - 1) / kThreads; }
constexpr int kPer1 = per_thread(5), kPer2 = per_thread(10), kPer3 = per_thread(kLocalCams);
__device__ __forceinline__ void fill_element_list(LocalState& S, int ncv, int tid) {
  const int total = ncv * (ncv + 1) / 2 * 36;
  for (int k = 0; k < kPer3; ++k) {
    const int idx = tid + k * kThreads;
    const int b = idx / 36, el = idx - b * 36;
    int bj = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
    while (bj * (bj + 1) / 2 > b) --bj;
    while ((bj + 1) * (bj + 2) / 2 <= b) ++bj;
    const int bi = b - bj * (bj + 1) / 2;
    const int ra = el / 6, cb = el - ra * 6;
    const bool take = idx < total && !(bi == bj && cb < ra);
    S.rc[idx] = take ? (int16_t)((bi * 6 + ra) << 8 | (bj * 6 + cb)) : (int16_t)-1;  // row <= column < 96
  }
}
template <int kPer>
__device__ __forceinline__ void gather_system(LocalState& S, DenseSolveLds& D, const double* acc, int tid) {
  double v[kPer];
  int rc[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    rc[k] = (int)S.rc[tid + k * kThreads];
    v[k] = rc[k] >= 0 ? acc[(rc[k] >> 8) * kLocalN + (rc[k] & 0xff)] : 0.0;
  }
  __syncthreads();  // S.v2 is written
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    if (rc[k] < 0) continue;
    const int R = rc[k] >> 8, C = rc[k] & 0xff;
    const double x = R == C ? v[k] + S.v2[R] : v[k];
    const int ti = C >> 5, tj = R >> 5;
    tile_at(D, ti, tj)[(C & 31) * kTLd + (R & 31)] = x;
    if (ti == tj && R != C) tile_at(D, ti, tj)[(R & 31) * kTLd + (C & 31)] = x;
  }
}

// (S + D / radius) y = W V^-1 g_p - g_c, all of it in LDS; acc: this iteration's accumulator (complete: behind barrier 1)
__device__ __forceinline__ void local_dense_solve(LocalState& S, DenseSolveLds& D, const double* acc, int n, double lm_radius, double min_diag,
                                                  double max_diag, long long* clk) {
  const int tid = thread_index<true>(), lane = tid & 63, wave = tid >> 6;
  const int mi = wave & 1, ni = wave >> 1;
  const int nt = (n + kTile - 1) / kTile;
  const double* gc = acc + kLocalN * kLocalN;
  const double* wv = gc + kLocalN;
  const double* dU = wv + kLocalN;
  long long c0 = 0, c1 = 0;
  if (clk) c0 = wall_clock64();
  // padding rows / columns of the tiles: identity
  for (int e = tid; e < nt * (nt + 1) / 2 * kTile * kTile; e += kThreads) {
    const int tile = e >> 10, r = (e >> 5) & 31, c = e & 31;
    const int ti = tile == 0 ? 0 : (tile < 3 ? 1 : 2), tj = tile - ti * (ti + 1) / 2;
    const int R = ti * kTile + r, C = tj * kTile + c;
    if (R >= n || C >= n) D.T[tile][r * kTLd + c] = R == C ? 1.0 : 0.0;
  }
  for (int c = tid; c < nt * kTile; c += kThreads) {
    const double g = c < n ? gc[c] : 0.0, w = c < n ? wv[c] : 0.0;
    S.rhs[c] = w - g; S.gcv[c] = g;
    S.v2[c] = c < n ? fmin(fmax(dU[c], min_diag), max_diag) / lm_radius : 0.0;  // the damping of column c
  }
  // the accumulator holds the upper block triangle: every element is requested once — all of a thread's loads are in flight
  // together, they come from beyond the L2 — and lands in the lower tile triangle (and mirrored inside a diagonal tile)
  if (nt == 1) gather_system<kPer1>(S, D, acc, tid);
  else if (nt == 2) gather_system<kPer2>(S, D, acc, tid);
  else gather_system<kPer3>(S, D, acc, tid);
  if (tid == 0) S.fail = 0;
  __syncthreads();
  if (clk) { c1 = wall_clock64(); clk[7] += c1 - c0; c0 = c1; }

  for (int j = 0; j < nt; ++j) {
    // stacked instances of tile column j, one per wave: 0 identity -> L(j,j)^-T, 1 right-hand side -> z_j, 2.. the tiles below
    const int ninst = 2 + (nt - 1 - j);
    if (wave < ninst) {
      double a[kTile];
      const int row = lane & 31;
      if (lane < kTile) {
        const double2* src = reinterpret_cast<const double2*>(tile_at(D, j, j) + row * kTLd);
#pragma unroll
        for (int c = 0; c < kTile; c += 2) { const double2 x = src[c >> 1]; a[c] = x.x; a[c + 1] = x.y; }
      } else if (wave == 0) {
#pragma unroll
        for (int c = 0; c < kTile; ++c) a[c] = c == row ? 1.0 : 0.0;
      } else if (wave == 1) {
#pragma unroll
        for (int c = 0; c < kTile; ++c) a[c] = row == 0 ? S.rhs[j * kTile + c] : 0.0;
      } else {
        const double2* src = reinterpret_cast<const double2*>(tile_at(D, j + wave - 1, j) + row * kTLd);
#pragma unroll
        for (int c = 0; c < kTile; c += 2) { const double2 x = src[c >> 1]; a[c] = x.x; a[c + 1] = x.y; }
      }
      bool ok = true;
      const int width = j == nt - 1 ? n - j * kTile : kTile;  // real columns of this tile column; whole panels of kSP
      if (width > 24) stacked_panel<0, 32>(a, lane, D.P[wave], ok);
      else if (width > 16) stacked_panel<0, 24>(a, lane, D.P[wave], ok);
      else if (width > 8) stacked_panel<0, 16>(a, lane, D.P[wave], ok);
      else stacked_panel<0, 8>(a, lane, D.P[wave], ok);
      if (lane >= kTile) {
        if (wave == 1) {
          if (row == 0) {
#pragma unroll
            for (int c = 0; c < kTile; ++c) S.z[j * kTile + c] = a[c];
          }
        } else {
          double2* dst = reinterpret_cast<double2*>((wave == 0 ? D.Linv[j] : tile_at(D, j + wave - 1, j)) + row * kTLd);
#pragma unroll
          for (int c = 0; c < kTile; c += 2) dst[c >> 1] = make_double2(a[c], a[c + 1]);
        }
      }
      if (wave == 0 && lane == 0 && !ok) S.fail = 1;
    }
    if (clk) { c1 = wall_clock64(); clk[8] += c1 - c0; c0 = c1; }
    __syncthreads();
    if (clk) { c1 = wall_clock64(); clk[9] += c1 - c0; c0 = c1; }
    // trailing tiles: T(i,k) -= L(i,j) L(k,j)^T, one quadrant per wave; right-hand side: rhs_k -= z_j L(k,j)^T
    for (int k = j + 1; k < nt; ++k)
      for (int i = k; i < nt; ++i) {
        double* Cik = tile_at(D, i, k);
        v4d c4;
        quad_load(Cik, kTLd, lane, mi, ni, c4);
        double am[8], bn[8];
        quad_operand_ld(tile_at(D, i, j), lane, mi, am);
        quad_operand_ld(tile_at(D, k, j), lane, ni, bn);
        quad_gemm_sub(am, bn, c4);
        quad_store(Cik, kTLd, lane, mi, ni, c4);
      }
    {
      const int k = j + 1 + (tid >> 5), c = tid & 31;
      if (k < nt) {
        const double* Lkj = tile_at(D, k, j) + c * kTLd;
        double s = 0.0;
#pragma unroll 8
        for (int m = 0; m < kTile; ++m) s += S.z[j * kTile + m] * Lkj[m];
        S.rhs[k * kTile + c] -= s;
      }
    }
    __syncthreads();
    if (clk) { c1 = wall_clock64(); clk[10] += c1 - c0; c0 = c1; }
  }
  // L^T y = z by tile columns, last first: y_j = L(j,j)^-T (z_j - sum_{i>j} L(i,j)^T y_i); eight partial sums per entry
  for (int j = nt - 1; j >= 0; --j) {
    const int c = tid & 31, part = tid >> 5;
    {
      double s = 0.0;
      for (int i = j + 1; i < nt; ++i) {
        const double* Lij = tile_at(D, i, j);
#pragma unroll
        for (int r = 4 * part; r < 4 * part + 4; ++r) s += Lij[r * kTLd + c] * S.y[i * kTile + r];
      }
      S.vp[part * kTile + c] = s;
    }
    __syncthreads();
    if (tid < kTile) {
      double s = S.z[j * kTile + tid];
#pragma unroll
      for (int p = 0; p < 8; ++p) s -= S.vp[p * kTile + tid];
      S.v[tid] = s;
    }
    __syncthreads();
    {
      const double* Li = D.Linv[j] + c * kTLd;
      double s = 0.0;
#pragma unroll
      for (int q = 4 * part; q < 4 * part + 4; ++q) s += Li[q] * S.v[q];
      S.vp[part * kTile + c] = s;
    }
    __syncthreads();
    if (tid < kTile) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < 8; ++p) s += S.vp[p * kTile + tid];
      S.y[j * kTile + tid] = s;
    }
    __syncthreads();
  }
  if (clk) { c1 = wall_clock64(); clk[11] += c1 - c0; }
}

}  // namespace

__global__ __launch_bounds__(kThreads, 1) void k_local_lm(LocalArgs G) {
  __shared__ PhaseLds U;
  __shared__ LocalState S;
  const SweepArgs& A = G.A;
  const int tid = threadIdx.x, cix = blockIdx.x, nwg = gridDim.x;
  const int ncv = G.ncv, n = 6 * ncv;
  const ChunkHdr H = A.chunks[cix];
  const bool writer = cix == 0;  // the workgroup that reports: traces, the log, the final cameras

  // ---- the workgroup's own copy of the control block and of the variable cameras ---------------------------------------------------
  if (tid == 0) { S.L = *static_cast<const LmHead*>(G.ctl); S.epoch = 0; S.fail = 0; }
  for (int i = tid; i < G.nc; i += kThreads) {
    const int slot = A.cam_slot[i];
    if (slot >= 0 && slot < kLocalCams) S.cam_of_slot[slot] = i;
  }
  __syncthreads();
  for (int e = tid; e < ncv * kCamRec; e += kThreads) {
    const int slot = e / kCamRec, k = e - slot * kCamRec;
    S.tab[e] = A.camtab[(size_t)S.cam_of_slot[slot] * kCamRec + k];
  }
  for (int e = tid; e < ncv * 4; e += kThreads) S.q[e] = G.q[(size_t)S.cam_of_slot[e >> 2] * 4 + (e & 3)];
  for (int e = tid; e < ncv * 3; e += kThreads) S.t[e] = G.t[(size_t)S.cam_of_slot[e / 3] * 3 + e % 3];
  for (int e = tid; e < ncv * 6; e += kThreads) S.cs[e] = G.cs[(size_t)S.cam_of_slot[e / 6] * 6 + e % 6];
  fill_element_list(S, ncv, tid);
  // the norm of the variable state (Ceres' x_norm of the parameter tolerance): this chunk's variable landmarks, summed over the
  // chunks behind a first grid barrier, and the variable cameras
  {
    double v = 0.0;
    for (int i = tid; i < H.npt; i += kThreads) {
      const int pix = H.pt0 + i;
      if (A.pt_kv[pix] != 0xffff) v += A.pts[3 * pix] * A.pts[3 * pix] + A.pts[3 * pix + 1] * A.pts[3 * pix + 1] + A.pts[3 * pix + 2] * A.pts[3 * pix + 2];
    }
    v = wave_sum(v);
    if ((tid & 63) == 0) S.red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) A.part2[(size_t)cix * 8 + 5] = (S.red[0] + S.red[1]) + (S.red[2] + S.red[3]);
  }
  if (!grid_barrier(G.bar, S, nwg)) return;
  {
    double v = 0.0;
    for (int r = tid; r < G.nchunks; r += kThreads) v += A.part2[(size_t)r * 8 + 5];
    v = wave_sum(v);
    if ((tid & 63) == 0) S.red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      double xn = (S.red[0] + S.red[1]) + (S.red[2] + S.red[3]);
      for (int e = 0; e < ncv * 4; ++e) xn += S.q[e] * S.q[e];
      for (int e = 0; e < ncv * 3; ++e) xn += S.t[e] * S.t[e];
      S.L.x_norm = sqrt(xn);
      S.L.fixed_cost = G.fixed_parts[0] + G.fixed_parts[1];
    }
    __syncthreads();
  }

  long long tk0 = 0;
  for (int it = 0;; ++it) {
    if (S.L.term != kLmRunning) break;  // (workgroup-uniform: every thread reads the same LDS word behind a barrier)
    const int par = it & 1;
    double* acc = G.acc[par];
    const double lm_radius = S.L.radius;
    if (writer && tid == 0) tk0 = wall_clock64();
    long long tka = 0;
    if ((A.dbg & 64) && tid == 0) tka = wall_clock64();

    // ---- A: track sweep of the chunk, slab into the accumulator ---------------------------------------------------------------------
    dense_sweep_chunk<true>(A, cix, lm_radius, S.tab, U.sweep);
    __syncthreads();  // the slab's stores (this workgroup's own) are visible to all its threads
    {
      const double* slab = A.slab + (size_t)H.slab0 * 18;
      const int ncam = H.ncam, nb = ncam * (ncam + 1) / 2;
      for (int idx = tid; idx < nb * 36; idx += kThreads) {
        const int b = idx / 36, el = idx - b * 36;
        int cj = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
        while (cj * (cj + 1) / 2 > b) --cj;
        while ((cj + 1) * (cj + 2) / 2 <= b) ++cj;
        const int ci = b - cj * (cj + 1) / 2;
        const int ra = el / 6, cb = el - ra * 6;
        if (ci == cj && cb < ra) continue;  // diagonal blocks carry their upper triangle
        const double v = slab[idx];
        if (v != 0.0) atomicAdd(&acc[(U.sweep.slot[ci] * 6 + ra) * kLocalN + U.sweep.slot[cj] * 6 + cb], v);
      }
      for (int idx = tid; idx < ncam * 18; idx += kThreads) {
        const int lc = idx / 18, k = idx - lc * 18;
        const double v = slab[nb * 36 + idx];
        if (v != 0.0) atomicAdd(&acc[kLocalN * kLocalN + (k / 6) * kLocalN + U.sweep.slot[lc] * 6 + k % 6], v);
      }
    }
    long long tkb = 0;
    if (writer && tid == 0) tkb = wall_clock64();
    if ((A.dbg & 64) && tid == 0) A.part[(size_t)cix * 4 + 3] = (double)(wall_clock64() - tka);  // this chunk's sweep + flush (diagnostics)
    if (!grid_barrier(G.bar, S, nwg)) return;
    long long tk1 = 0;
    if (writer && tid == 0) tk1 = wall_clock64();

    // ---- B: the reduced system, every workgroup for itself ------------------------------------------------------------------------------
    local_dense_solve(S, U.dense, acc, n, lm_radius, A.min_diag, A.max_diag, (writer && tid == 0 && (A.dbg & 64)) ? G.clk : nullptr);

    // ---- C: candidate cameras (camera_candidate: k_cam_update's arithmetic), one thread per variable camera ---------------------------------------------
    if (tid < ncv) {
      const int sl = tid;
      const double qq[4] = {S.q[4 * sl], S.q[4 * sl + 1], S.q[4 * sl + 2], S.q[4 * sl + 3]};
      const double tt[3] = {S.t[3 * sl], S.t[3 * sl + 1], S.t[3 * sl + 2]};
      double qn[4], tn[3], step = 0.0, xn = 0.0, gmax = 0.0;
      camera_candidate(qq, tt, &S.cs[6 * sl], &S.y[6 * sl], &S.gcv[6 * sl], qn, tn, step, xn, gmax);
      S.camred[3 * sl] = step; S.camred[3 * sl + 1] = xn; S.camred[3 * sl + 2] = gmax;
      for (int k = 0; k < 4; ++k) S.q2[4 * sl + k] = qn[k];
      for (int k = 0; k < 3; ++k) S.t2[3 * sl + k] = tn[k];
      double* o = &S.tab2[sl * kCamRec];
      quat_to_R(qn, o);
      o[9] = tn[0]; o[10] = tn[1]; o[11] = tn[2];
      for (int k = 12; k < kCamRec; ++k) o[k] = S.tab[sl * kCamRec + k];  // intrinsics, column scales, padding
    }
    __syncthreads();
    long long tk2 = 0;
    if (writer && tid == 0) tk2 = wall_clock64();

    // ---- D: update sweep of the chunk; the other accumulator starts the next iteration from zero --------------------------------------
    update_sweep_chunk<true>(A, cix, lm_radius, S.tab, S.tab2, S.y, U.upd, CamUpdArgs{}, false);
    {
      double* other = G.acc[par ^ 1];
      for (int e = cix * kThreads + tid; e < kLocalAccDoubles; e += nwg * kThreads) other[e] = 0.0;
    }
    long long tkc = 0;
    if (writer && tid == 0) tkc = wall_clock64();
    if (!grid_barrier(G.bar, S, nwg)) return;
    long long tkd = 0;
    if (writer && tid == 0) tkd = wall_clock64();

    // ---- E: the iteration's scalars (k_lm_reduce_decide's order) and the decision, the same in every workgroup -----------------------
    {
      // eight columns (0-2 track sweep: cost, invalid count, gradient maximum; 3-7 update sweep): every thread brings one row (all
      // loads of the workgroup in flight together: they come from beyond the L2) into LDS — the phase storage is free here —, then one
      // wave adds them, eight lanes per column, in a fixed order: cheaper than eight wave reductions
      double* rows = reinterpret_cast<double*>(&U);
      for (int r = tid; r < G.nchunks; r += kThreads) {
        const double* a = A.part + (size_t)r * 4;
        const double* b = A.part2 + (size_t)r * 8;
        const double a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4];
        double* o = rows + (size_t)r * 8;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = b0; o[4] = b1; o[5] = b2; o[6] = b3; o[7] = b4;
      }
      __syncthreads();
      if (tid < 64) {  // lane = column + 8 * part: eight row classes per column, then three shuffle steps across the parts
        const int c = tid & 7, part = tid >> 3;
        const bool is_max = c == 2;
        double p0 = 0.0, p1 = 0.0;
        int r = part;
        for (; r + 8 < G.nchunks; r += 16) {
          const double x0 = rows[r * 8 + c], x1 = rows[(r + 8) * 8 + c];
          p0 = is_max ? fmax(p0, x0) : p0 + x0;
          p1 = is_max ? fmax(p1, x1) : p1 + x1;
        }
        if (r < G.nchunks) { const double x0 = rows[r * 8 + c]; p0 = is_max ? fmax(p0, x0) : p0 + x0; }
        double v = is_max ? fmax(p0, p1) : p0 + p1;
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) {
          const double o = __shfl_down(v, off, 64);
          v = is_max ? fmax(v, o) : v + o;
        }
        if (tid < 8) S.out[tid] = v;
      }
      __syncthreads();
      if (tid == 0) {
        double sc[U_COUNT];
#pragma unroll
        for (int c = 0; c < U_COUNT; ++c) sc[c] = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) sc[c < 3 ? U_X_COST + c : c - 3] = S.out[c];
        double step = 0.0, xn = 0.0, gmax = 0.0;
        for (int sl = 0; sl < ncv; ++sl) { step += S.camred[3 * sl]; xn += S.camred[3 * sl + 1]; gmax = fmax(gmax, S.camred[3 * sl + 2]); }
        sc[U_STEP_SQ_CAMS] = step; sc[U_XN_SQ_CAMS] = xn; sc[U_GMAX_CAMS] = gmax;
        sc[U_CHOL_FAIL] = S.fail ? 1.0 : 0.0;
        LmHead L = S.L;
        lm_decide_logic(L, writer ? G.ctl : nullptr, sc, G.o);
        S.L = L;
        if (writer && G.log) G.log[it] = L;
      }
      __syncthreads();
    }
    if (S.L.accepted) {
      for (int e = tid; e < ncv * kCamRec; e += kThreads) S.tab[e] = S.tab2[e];
      for (int e = tid; e < ncv * 4; e += kThreads) S.q[e] = S.q2[e];
      for (int e = tid; e < ncv * 3; e += kThreads) S.t[e] = S.t2[e];
      for (int e = tid; e < 3 * H.npt; e += kThreads) G.pts[(size_t)3 * H.pt0 + e] = A.pts2[(size_t)3 * H.pt0 + e];
    }
    __syncthreads();
    if (writer && tid == 0) {
      const long long tk3 = wall_clock64();
      G.clk[0] += tkb - tk0; G.clk[1] += tk1 - tkb; G.clk[2] += tk2 - tk1; G.clk[3] += tkc - tk2; G.clk[4] += tkd - tkc; G.clk[5] += tk3 - tkd;
      G.clk[6] += 1;
    }
  }

  // ---- the cameras and the control block go back -----------------------------------------------------------------------------------------
  if (writer) {
    for (int e = tid; e < ncv * kCamRec; e += kThreads) G.camtab[(size_t)S.cam_of_slot[e / kCamRec] * kCamRec + e % kCamRec] = S.tab[e];
    for (int e = tid; e < ncv * 4; e += kThreads) G.q[(size_t)S.cam_of_slot[e >> 2] * 4 + (e & 3)] = S.q[e];
    for (int e = tid; e < ncv * 3; e += kThreads) G.t[(size_t)S.cam_of_slot[e / 3] * 3 + e % 3] = S.t[e];
    if (tid == 0) *static_cast<LmHead*>(G.ctl) = S.L;
  }
}

int local_lm_max_chunks(int device) {
  int coop = 0;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess || !coop) { (void)hipGetLastError(); return 0; }
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_local_lm, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return per_cu * cus;
}

int launch_local_lm(const LocalArgs& a, hipStream_t s) {
  LocalArgs copy = a;
  void* params[] = {&copy};
  return (int)hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_local_lm), dim3((unsigned)a.nchunks), dim3(kThreads), params, 0, s);
}

}  // namespace mpsfm
