// Track-sweep kernels of the bundle-adjustment hot path (gfx950).
//
// Replaces what Ceres does per LM iteration inside pyceres.solve for the problem assembled by
// reference mpsfm/sfm/mapper/bundle_adjustment.py:67-185:
//   k_track_sweep   residual + analytic Jacobian evaluation of every (camera, landmark) record,
//                   J^T J block accumulation and the landmark Schur reduce.  One workgroup owns a
//                   chunk of consecutive landmarks; W blocks never leave LDS; the chunk's
//                   contributions to the reduced camera system are summed in an LDS tile of
//                   6x6 blocks (ds_add_f64) and flushed once with global_atomic_add_f64.
//   k_update_sweep  back-substitution of the landmark steps, model cost change, candidate
//                   landmarks and the candidate cost in one pass over the same chunks.
#include "common.h"
#include "sweep_common.h"
#include "sweep_update_body.h"
#include "lm_decide.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>

namespace mpsfm {




// Track sweep.  One workgroup per chunk of landmarks:
//   P1  one thread per merged record: residuals, analytic Jacobians, robust weights; V_p / g_p
//       (ds_add_f64 per landmark), U_c / g_c / diag U (ds_add_f64 per local camera), W = Jc^T Jp to LDS
//   P2  one thread per landmark: (V + D)^-1 and (V + D)^-1 g_p
//   P3a rhs part  sum_p W Vinv g_p  per camera
//   P3b Schur pairs, block-major: the chunk's pairs are pre-sorted (at problem creation) by the 6x6
//       destination block of S; six lanes own one block (a row each), sum its pairs in registers
//       and issue ONE global_atomic_add_f64 per block element.  No LDS atomics on this path.
//   P4  flush the per-camera LDS accumulators, write the chunk partials
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_track_sweep(SweepArgs A) {
  if (lm_over(A.ctl)) return;
  const double lm_radius = A.ctl ? lm_radius_of(A.ctl) : A.radius;
  __shared__ __attribute__((aligned(16))) double s_W[kObsMax * kWStride];
  __shared__ __attribute__((aligned(16))) double s_V[kPtsMax * 6];
  __shared__ double s_g[kPtsMax * 3];
  // per-camera accumulators: neighbouring lanes are the records of ONE landmark, so a wave adds ~13 records to every camera of
  // the chunk with one instruction — kCamCopies copies (chosen by the landmark) cut that same-address serialisation; P4 sums them
  constexpr int kCamCopies = MPSFM_CAM_COPIES, kUCopy = kTileCams * 21 + 1, kGCopy = kTileCams * 6 + 1;
  __shared__ double s_U[kCamCopies * kUCopy];
  __shared__ double s_gc[kCamCopies * kGCopy];
  __shared__ double s_wv[kCamCopies * kGCopy];
  __shared__ double s_du[kTileCams * 6];
  __shared__ int32_t s_slot[kLocalCamsMax];
  __shared__ uint8_t s_const[kPtsMax];
  __shared__ double s_red[3 * (kThreads / 64)];
  __shared__ double s_stage[(kThreads / kPairGroup) * 36];
  __shared__ uint32_t s_ents[kEntStage];

  const int tid = threadIdx.x;
  const int cix = blockIdx.x + A.chunk0;  // FULL mode: the launch covers the chunks the dense sweep does not take
  const ChunkHdr H = A.chunks[cix];
  const int nrec = H.nrec, npt = H.npt, ncam = H.ncam;

  // ---- P0: clear accumulators, stage chunk tables -----------------------------------------
  for (int i = tid; i < kPtsMax * 6; i += kThreads) s_V[i] = 0.0;
  for (int i = tid; i < kPtsMax * 3; i += kThreads) s_g[i] = 0.0;
  for (int i = tid; i < kCamCopies * kUCopy; i += kThreads) s_U[i] = 0.0;
  for (int i = tid; i < kCamCopies * kGCopy; i += kThreads) { s_gc[i] = 0.0; s_wv[i] = 0.0; }
  if (tid < kTileCams * 6) s_du[tid] = 0.0;
  if (tid < ncam) s_slot[tid] = A.chunk_cams[H.cam0 + tid];
  // what P2 needs of this thread's landmark is requested here, so that the loads are back when P1 is done
  uint16_t my_kv = 0xffff;
  double my_ps[3] = {1.0, 1.0, 1.0};
  if (MODE == MODE_FULL && tid < npt) {
    my_kv = A.pt_kv[H.pt0 + tid];
    my_ps[0] = A.ps[3 * (H.pt0 + tid)]; my_ps[1] = A.ps[3 * (H.pt0 + tid) + 1]; my_ps[2] = A.ps[3 * (H.pt0 + tid) + 2];
    s_const[tid] = my_kv == 0xffff;  // constant landmark: no Schur products (P3a)
  }
  const bool ents_in_lds = (H.nent <= kEntStage);
  if (MODE == MODE_FULL && ents_in_lds) {
    for (int i = tid; i < H.nent; i += kThreads) s_ents[i] = A.ents[H.ent0 + i];
  }
  __syncthreads();
  if (A.dbg & 8) return;  // ablation: stop after P0

  // ---- P1: per-record residual / Jacobian, J^T J blocks ------------------------------------
  double my_cost = 0.0;
  int my_bad = 0;
  uint32_t my_meta = 0;
  // W = Jc^T Jp (6x3) of this thread's record stays in registers until P3a turns it into Z (two workgroups per CU by LDS:
  // registers are not what limits occupancy)
  double my_w[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) my_w[i] = 0.0;
  if (tid < nrec) {
    const int rix = H.rec0 + tid;
    const uint32_t meta = A.rec_meta[rix];
    my_meta = meta;
    const int cam = A.rec_cam[rix];
    const int lcam = meta & 0xff;
    const int lpt = (meta >> 8) & 0xff;
    const double2 xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    double d = 1.0, m = 0.0, a = 1.0;
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    const int pix = H.pt0 + lpt;
    const double X[3] = {A.pts[3 * pix], A.pts[3 * pix + 1], A.pts[3 * pix + 2]};
    const double psc[3] = {A.ps[3 * pix], A.ps[3 * pix + 1], A.ps[3 * pix + 2]};
    RecLin L;
    linearize_record(A.camtab + (size_t)cam * kCamRec, X, psc, meta, xy.x, xy.y, d, m, a, A.loss, L);
    my_cost = L.cost;
    my_bad = L.ok ? 0 : 1;
    if (L.ok) {
      // landmark block
      if (psc[0] != 0.0) {
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double j0 = L.Jp[3 * r], j1 = L.Jp[3 * r + 1], j2 = L.Jp[3 * r + 2];
          V[0] += j0 * j0; V[1] += j0 * j1; V[2] += j0 * j2; V[3] += j1 * j1; V[4] += j1 * j2; V[5] += j2 * j2;
          g[0] += j0 * L.r[r]; g[1] += j1 * L.r[r]; g[2] += j2 * L.r[r];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&s_V[lpt * 6 + k], V[k]);
        if (MODE == MODE_FULL) {
#pragma unroll
          for (int k = 0; k < 3; ++k) atomicAdd(&s_g[lpt * 3 + k], g[k]);
        }
      }
      if (lcam != (int)kLcamConst && !(A.dbg & 1)) {
        const int slot = s_slot[lcam];
        // diag(U): in the full sweep the LDS copy comes for free from the diagonal of the U tile (P4)
        if (MODE == MODE_DIAG || lcam >= kTileCams) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double du = L.Jc[i] * L.Jc[i] + L.Jc[6 + i] * L.Jc[6 + i] + L.Jc[12 + i] * L.Jc[12 + i];
            if (lcam < kTileCams) atomicAdd(&s_du[lcam * 6 + i], du);
            else atomicAdd(&A.diagU[(size_t)slot * 6 + i], du);
          }
        }
        if (MODE == MODE_FULL) {
          // g_c and the upper triangle of U_c (21 values, packed row-major a <= b)
          double* gS = &A.Sblk[sky_block(A.sky, slot, slot) * 36];
          const int copy = lpt % kCamCopies;
          int u = 0;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double gci = L.Jc[i] * L.r[0] + L.Jc[6 + i] * L.r[1] + L.Jc[12 + i] * L.r[2];
            if (lcam < kTileCams) atomicAdd(&s_gc[copy * kGCopy + lcam * 6 + i], gci);
            else atomicAdd(&A.gc[(size_t)slot * 6 + i], gci);
#pragma unroll
            for (int j = i; j < 6; ++j, ++u) {
              const double uij = L.Jc[i] * L.Jc[j] + L.Jc[6 + i] * L.Jc[6 + j] + L.Jc[12 + i] * L.Jc[12 + j];
              if (lcam < kTileCams) atomicAdd(&s_U[copy * kUCopy + lcam * 21 + u], uij);
              else atomicAdd(&gS[i * 6 + j], uij);
            }
          }
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              my_w[i * 3 + j] = L.Jc[i] * L.Jp[j] + L.Jc[6 + i] * L.Jp[3 + j] + L.Jc[12 + i] * L.Jp[6 + j];
        }
      }
    }
  }
  __syncthreads();
  if (A.dbg & 16) return;  // ablation: stop after P1

  // ---- P2: per-landmark (V + D)^-1 ------------------------------------------------------------
  double my_gmax = 0.0;
  if (tid < npt) {
    const int pix = H.pt0 + tid;
    if (MODE == MODE_DIAG) {
      A.diagV[3 * pix] = s_V[tid * 6];
      A.diagV[3 * pix + 1] = s_V[tid * 6 + 3];
      A.diagV[3 * pix + 2] = s_V[tid * 6 + 5];
    } else if (my_kv != 0xffff) {
      // variable landmark (kv == 0xffff marks a constant one)
      double V[6], Vi[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) V[k] = s_V[tid * 6 + k];
      V[0] += fmin(fmax(V[0], A.min_diag), A.max_diag) / lm_radius;
      V[3] += fmin(fmax(V[3], A.min_diag), A.max_diag) / lm_radius;
      V[5] += fmin(fmax(V[5], A.min_diag), A.max_diag) / lm_radius;
      // Vi <- F = L^-1 of V + D = L L^T: the Schur products become Z Z'^T with Z = W F^T (see spd3_inv_factor)
      if (!spd3_inv_factor(V, Vi)) {
        my_bad = 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) Vi[k] = 0.0;
      }
      const double g0 = s_g[tid * 3], g1 = s_g[tid * 3 + 1], g2 = s_g[tid * 3 + 2];
#pragma unroll
      for (int k = 0; k < 6; ++k) s_V[tid * 6 + k] = Vi[k];
      s_g[tid * 3] = Vi[0] * g0; s_g[tid * 3 + 1] = Vi[1] * g0 + Vi[2] * g1; s_g[tid * 3 + 2] = Vi[3] * g0 + Vi[4] * g1 + Vi[5] * g2;  // F g
      my_gmax = fmax(fabs(g0 / my_ps[0]), fmax(fabs(g1 / my_ps[1]), fabs(g2 / my_ps[2])));
    }
  }
  if (MODE == MODE_FULL) {
    __syncthreads();

    // ---- P3a: rhs part  sum_p W Vinv g_p ------------------------------------------------------
    if (tid < nrec) {
      const int lcam = my_meta & 0xff;
      const int lpt = (my_meta >> 8) & 0xff;
      if (lcam != (int)kLcamConst && !s_const[lpt]) {
        double* w = &s_W[tid * kWStride];
        const double v0 = s_g[lpt * 3], v1 = s_g[lpt * 3 + 1], v2 = s_g[lpt * 3 + 2];
        const double f00 = s_V[lpt * 6], f10 = s_V[lpt * 6 + 1], f11 = s_V[lpt * 6 + 2], f20 = s_V[lpt * 6 + 3], f21 = s_V[lpt * 6 + 4],
                     f22 = s_V[lpt * 6 + 5];
        const int slot = s_slot[lcam];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          // W -> Z = W F^T, into the record's row block in LDS
          const double w0 = my_w[i * 3], w1 = my_w[i * 3 + 1], w2 = my_w[i * 3 + 2];
          const double z0 = w0 * f00, z1 = w0 * f10 + w1 * f11, z2 = w0 * f20 + w1 * f21 + w2 * f22;
          w[i * 3] = z0; w[i * 3 + 1] = z1; w[i * 3 + 2] = z2;
          const double x = z0 * v0 + z1 * v1 + z2 * v2;
          if (lcam < kTileCams) atomicAdd(&s_wv[(lpt % kCamCopies) * kGCopy + lcam * 6 + i], x);
          else atomicAdd(&A.wv[(size_t)slot * 6 + i], x);
        }
      }
    }

    __syncthreads();  // the rows of s_W now hold Z


    // ---- P3b: Schur pairs, block-major:  S[ci,cj] -= sum_pairs Z_i Z_j^T  (= W_i (V+D)^-1 W_j^T) ---------
    // Rounds of kGroups blocks: six lanes sum one block in registers, park it in LDS, then the whole
    // workgroup flushes the round with lanes running along the 36 contiguous doubles of a block
    // (the access shape global atomics want).
    {
      constexpr int kGroups = kThreads / kPairGroup;
      const int grp = tid / kPairGroup, row = tid - grp * kPairGroup;
      const int nblk = (A.dbg & 2) ? 0 : H.nblk;
      for (int b0 = 0; b0 < nblk; b0 += kGroups) {
        const int b = b0 + grp;
        if (grp < kGroups && b < nblk) {
          // entry offsets are chunk-relative
          int e = A.blk_ent_start[H.blk0 + cix + b];
          const int e1 = A.blk_ent_start[H.blk0 + cix + b + 1];
          const uint32_t* ep = ents_in_lds ? s_ents : (A.ents + H.ent0);
          double acc[6] = {0, 0, 0, 0, 0, 0};
          uint32_t ent = (e < e1) ? ep[e] : 0u;
          while (e < e1) {
            const uint32_t cur = ent;
            ++e;
            if (e < e1) ent = ep[e];  // fetch the next pair while this one is processed
            const int ri = cur & 0xff, rj = (cur >> 8) & 0xff;
            const double* zi = &s_W[ri * kWStride + row * 3];
            // 16-byte LDS reads: Z_j (9 x b128) is shared by the six lanes of the group
            double wj[18];
            {
              const double2* w2 = reinterpret_cast<const double2*>(&s_W[rj * kWStride]);
#pragma unroll
              for (int k = 0; k < 9; ++k) { const double2 t = w2[k]; wj[2 * k] = t.x; wj[2 * k + 1] = t.y; }
            }
            const double y[3] = {zi[0], zi[1], zi[2]};
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) acc[bb] += y[0] * wj[bb * 3] + y[1] * wj[bb * 3 + 1] + y[2] * wj[bb * 3 + 2];
          }
#pragma unroll
          for (int bb = 0; bb < 6; ++bb) s_stage[grp * 36 + row * 6 + bb] = acc[bb];
        }
        __syncthreads();
        const int nround = min(kGroups, nblk - b0);
        if (!(A.dbg & 4)) {
          for (int idx = tid; idx < nround * 36; idx += kThreads) {
            const int g = idx / 36, el = idx - g * 36;
            const uint32_t desc = A.blk_desc[H.blk0 + b0 + g];
            // work items of one block sit next to each other: the first of a run flushes the run's sum,
            // so one wave-instruction never adds twice to the same address
            if (g > 0 && A.blk_desc[H.blk0 + b0 + g - 1] == desc) continue;
            double v = s_stage[idx];
            for (int g2 = g + 1; g2 < nround && A.blk_desc[H.blk0 + b0 + g2] == desc; ++g2) v += s_stage[g2 * 36 + el];
            const int li = desc & 0xff, lj = (desc >> 8) & 0xff;
            const int ra = el / 6, cb = el - ra * 6;
            if (v != 0.0 && (li != lj || cb >= ra))  // diagonal blocks keep their upper triangle only
              atomicAdd(&A.Sblk[sky_block(A.sky, s_slot[li], s_slot[lj]) * 36 + el], -v);
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();

  // ---- P4: flush the per-camera LDS accumulators, chunk partials ------------------------------
  if (MODE == MODE_FULL) {
    for (int idx = tid; idx < kTileCams * 21; idx += kThreads) {
      const int lc = idx / 21;
      double v = s_U[idx];
#pragma unroll
      for (int q = 1; q < kCamCopies; ++q) v += s_U[q * kUCopy + idx];
      if (lc < ncam && v != 0.0) {
        int u = idx - lc * 21, i = 0;
        while (u >= 6 - i) { u -= 6 - i; ++i; }  // packed (i, j >= i) -> i, j = i + u
        atomicAdd(&A.Sblk[sky_block(A.sky, s_slot[lc], s_slot[lc]) * 36 + i * 6 + i + u], v);
        if (u == 0) atomicAdd(&A.diagU[(size_t)s_slot[lc] * 6 + i], v);  // diagonal entry of U
      }
    }
  }
  if (tid < kTileCams * 6) {
    const int lc = tid / 6;
    if (lc < ncam) {
      const size_t o = (size_t)s_slot[lc] * 6 + (tid - lc * 6);
      if (MODE == MODE_DIAG && s_du[tid] != 0.0) atomicAdd(&A.diagU[o], s_du[tid]);
      if (MODE == MODE_FULL) {
        double g = s_gc[tid], w = s_wv[tid];
#pragma unroll
        for (int q = 1; q < kCamCopies; ++q) { g += s_gc[q * kGCopy + tid]; w += s_wv[q * kGCopy + tid]; }
        if (g != 0.0) atomicAdd(&A.gc[o], g);
        if (w != 0.0) atomicAdd(&A.wv[o], w);
      }
    }
  }
  if (MODE == MODE_FULL) {
    const double c = wave_sum(my_cost);
    const double b = wave_sum((double)my_bad);
    const double g = wave_max(my_gmax);
    const int w = tid >> 6;
    if ((tid & 63) == 0) { s_red[w] = c; s_red[4 + w] = b; s_red[8 + w] = g; }
    __syncthreads();
    if (tid == 0) {
      double* p = A.part + (size_t)cix * 4;
      p[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
      p[1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
      p[2] = fmax(fmax(s_red[8], s_red[9]), fmax(s_red[10], s_red[11]));
      p[3] = 0.0;
    }
  }
}

// -----------------------------------------------------------------------------------------------
// Update sweep: y_p = -(V+D)^-1 (g_p + W^T y_c), model cost change, candidate landmarks, candidate
// cost.  Recomputes the linearisation (cheaper than storing 240 B per record in HBM).
#ifndef MPSFM_UPD_OCC
#define MPSFM_UPD_OCC 4  // 102 registers without spills; LDS (12 KB) no longer limits
#endif
__global__ __launch_bounds__(kThreads, MPSFM_UPD_OCC) void k_update_sweep(SweepArgs A, CamUpdArgs U) {
  if (lm_over(A.ctl)) return;
  const double lm_radius = A.ctl ? lm_radius_of(A.ctl) : A.radius;
  __shared__ UpdLds S;
  if (U.fuse && blockIdx.x == 0) cam_update_all(U, A.yc, S.red);  // (the candidate rows of this chunk come next, like everywhere)
  update_sweep_chunk<false>(A, blockIdx.x, lm_radius, nullptr, nullptr, A.yc, S, U, U.fuse != 0);
}

// decode q in [0, k(k+1)/2) -> (i, j), i <= j < k, row-major upper triangle
__device__ __forceinline__ void tri_decode(int64_t q, int k, int& i, int& j) {
  const double b = 2.0 * k + 1.0;
  int64_t ii = (int64_t)((b - sqrt(b * b - 8.0 * (double)q)) * 0.5);
  if (ii < 0) ii = 0;
  if (ii > k - 1) ii = k - 1;
  while (ii > 0 && ii * k - (ii * (ii - 1)) / 2 > q) --ii;       // row start(i) = i*k - i(i-1)/2
  while ((ii + 1) * k - ((ii + 1) * ii) / 2 <= q) ++ii;
  i = (int)ii;
  j = (int)(q - (ii * k - (ii * (ii - 1)) / 2) + ii);
}

// block-wide sum of N doubles per thread; result valid in every thread
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* s_buf /* N * 4 */) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double w = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) s_buf[k * 4 + (threadIdx.x >> 6)] = w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = (s_buf[k * 4] + s_buf[k * 4 + 1]) + (s_buf[k * 4 + 2] + s_buf[k * 4 + 3]);
  __syncthreads();
}

// Track sweep of ONE landmark whose track is too long for a chunk: the workgroup strides over the
// records; W goes to an HBM scratch; U / g_c / Schur blocks are added to the reduced system directly.
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_long_track_sweep(SweepArgs A) {
  if (lm_over(A.ctl)) return;
  const double lm_radius = A.ctl ? lm_radius_of(A.ctl) : A.radius;
  __shared__ double s_buf[9 * 4];
  __shared__ double s_vi[9];
  const int tid = threadIdx.x;
  const LongHdr H = A.lhdr[blockIdx.x];
  const int pix = H.pt;
  const double X[3] = {A.pts[3 * pix], A.pts[3 * pix + 1], A.pts[3 * pix + 2]};
  const double psc[3] = {A.ps[3 * pix], A.ps[3 * pix + 1], A.ps[3 * pix + 2]};
  const bool pvar = A.pt_kv[pix] != 0xffff;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // V (6), g (3)
  double my_cost = 0.0, my_bad = 0.0;
  for (int r = tid; r < H.nrec; r += kThreads) {
    const int rix = H.rec0 + r;
    const uint32_t meta = A.rec_meta[rix];
    const int cam = A.rec_cam[rix];
    const int slot = A.cam_slot[cam];
    const double2 xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    double d = 1.0, m = 0.0, a = 1.0;
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    RecLin L;
    linearize_record(A.camtab + (size_t)cam * kCamRec, X, psc, meta, xy.x, xy.y, d, m, a, A.loss, L);
    my_cost += L.cost;
    if (!L.ok) { my_bad += 1.0; continue; }
    if (psc[0] != 0.0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double j0 = L.Jp[3 * q], j1 = L.Jp[3 * q + 1], j2 = L.Jp[3 * q + 2];
        acc[0] += j0 * j0; acc[1] += j0 * j1; acc[2] += j0 * j2; acc[3] += j1 * j1; acc[4] += j1 * j2; acc[5] += j2 * j2;
        acc[6] += j0 * L.r[q]; acc[7] += j1 * L.r[q]; acc[8] += j2 * L.r[q];
      }
    }
    if (slot >= 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        atomicAdd(&A.diagU[(size_t)slot * 6 + i], L.Jc[i] * L.Jc[i] + L.Jc[6 + i] * L.Jc[6 + i] + L.Jc[12 + i] * L.Jc[12 + i]);
      if (MODE == MODE_FULL) {
        double* gS = &A.Sblk[sky_block(A.sky, slot, slot) * 36];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          atomicAdd(&A.gc[(size_t)slot * 6 + i], L.Jc[i] * L.r[0] + L.Jc[6 + i] * L.r[1] + L.Jc[12 + i] * L.r[2]);
#pragma unroll
          for (int j = i; j < 6; ++j)
            atomicAdd(&gS[i * 6 + j], L.Jc[i] * L.Jc[j] + L.Jc[6 + i] * L.Jc[6 + j] + L.Jc[12 + i] * L.Jc[12 + j]);
        }
        double* w = A.wl + (size_t)(H.w0 + r) * 18;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) w[i * 3 + j] = L.Jc[i] * L.Jp[j] + L.Jc[6 + i] * L.Jp[3 + j] + L.Jc[12 + i] * L.Jp[6 + j];
      }
    }
  }
  block_sum<9>(acc, s_buf);
  if (MODE == MODE_DIAG) {
    if (tid == 0) { A.diagV[3 * pix] = acc[0]; A.diagV[3 * pix + 1] = acc[3]; A.diagV[3 * pix + 2] = acc[5]; }
    return;
  }
  double my_gmax = 0.0;
  if (tid == 0) {
    double Vi[6] = {0, 0, 0, 0, 0, 0}, vg[3] = {0, 0, 0};
    if (pvar) {
      double V[6] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]};
      V[0] += fmin(fmax(V[0], A.min_diag), A.max_diag) / lm_radius;
      V[3] += fmin(fmax(V[3], A.min_diag), A.max_diag) / lm_radius;
      V[5] += fmin(fmax(V[5], A.min_diag), A.max_diag) / lm_radius;
      if (!spd3_inverse(V, Vi)) { my_bad += 1.0; for (int k = 0; k < 6; ++k) Vi[k] = 0.0; }
      sym3_mul(Vi, acc[6], acc[7], acc[8], vg);
      my_gmax = fmax(fabs(acc[6] / psc[0]), fmax(fabs(acc[7] / psc[1]), fabs(acc[8] / psc[2])));
    }
    for (int k = 0; k < 6; ++k) s_vi[k] = Vi[k];
    for (int k = 0; k < 3; ++k) s_vi[6 + k] = vg[k];
  }
  __syncthreads();
  if (pvar) {
    double Vi[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Vi[k] = s_vi[k];
    const double v0 = s_vi[6], v1 = s_vi[7], v2 = s_vi[8];
    __threadfence();  // the W rows written above are read back by other threads of this workgroup
    __syncthreads();
    for (int r = tid; r < H.kv; r += kThreads) {
      const double* w = A.wl + (size_t)(H.w0 + r) * 18;
      const int slot = A.cam_slot[A.rec_cam[H.rec0 + r]];
#pragma unroll
      for (int i = 0; i < 6; ++i) atomicAdd(&A.wv[(size_t)slot * 6 + i], w[i * 3] * v0 + w[i * 3 + 1] * v1 + w[i * 3 + 2] * v2);
    }
    const int64_t npairs = (int64_t)H.kv * (H.kv + 1) / 2;
    for (int64_t e = tid; e < npairs; e += kThreads) {
      int i, j;
      tri_decode(e, H.kv, i, j);
      const int si = A.cam_slot[A.rec_cam[H.rec0 + i]], sj = A.cam_slot[A.rec_cam[H.rec0 + j]];
      const double* wi = A.wl + (size_t)(H.w0 + i) * 18;
      const double* wj = A.wl + (size_t)(H.w0 + j) * 18;
      double Y[18], Wj[18];
#pragma unroll
      for (int a = 0; a < 6; ++a) sym3_mul(Vi, wi[a * 3], wi[a * 3 + 1], wi[a * 3 + 2], &Y[a * 3]);
#pragma unroll
      for (int k = 0; k < 18; ++k) Wj[k] = wj[k];
      double* dst = &A.Sblk[sky_block(A.sky, si, sj) * 36];
      if (si != sj) {
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b < 6; ++b)
            atomicAdd(&dst[a * 6 + b], -(Y[a * 3] * Wj[b * 3] + Y[a * 3 + 1] * Wj[b * 3 + 1] + Y[a * 3 + 2] * Wj[b * 3 + 2]));
      } else {
        double Yj[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) sym3_mul(Vi, Wj[a * 3], Wj[a * 3 + 1], Wj[a * 3 + 2], &Yj[a * 3]);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) {
            double v = Y[a * 3] * Wj[b * 3] + Y[a * 3 + 1] * Wj[b * 3 + 1] + Y[a * 3 + 2] * Wj[b * 3 + 2];
            if (i != j) v += Yj[a * 3] * wi[b * 3] + Yj[a * 3 + 1] * wi[b * 3 + 1] + Yj[a * 3 + 2] * wi[b * 3 + 2];
            atomicAdd(&dst[a * 6 + b], -v);
          }
      }
    }
  }
  double red[3] = {my_cost, my_bad, 0.0};
  block_sum<3>(red, s_buf);
  const double g = wave_max(my_gmax);  // only thread 0 (wave 0) holds a value
  if (tid == 0) {
    double* p = A.part + (size_t)(A.nchunks + blockIdx.x) * 4;
    p[0] = red[0]; p[1] = red[1]; p[2] = g; p[3] = 0.0;
  }
}

// Update sweep of one long-track landmark (see k_update_sweep).
__global__ __launch_bounds__(kThreads) void k_long_update_sweep(SweepArgs A) {
  if (lm_over(A.ctl)) return;
  const double lm_radius = A.ctl ? lm_radius_of(A.ctl) : A.radius;
  __shared__ double s_buf[9 * 4];
  __shared__ double s_b[6];
  const int tid = threadIdx.x;
  const LongHdr H = A.lhdr[blockIdx.x];
  const int pix = H.pt;
  const double X[3] = {A.pts[3 * pix], A.pts[3 * pix + 1], A.pts[3 * pix + 2]};
  const double psc[3] = {A.ps[3 * pix], A.ps[3 * pix + 1], A.ps[3 * pix + 2]};
  const bool pvar = A.pt_kv[pix] != 0xffff;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double bad = 0.0;
  for (int r = tid; r < H.nrec; r += kThreads) {
    const int rix = H.rec0 + r;
    const uint32_t meta = A.rec_meta[rix];
    const int cam = A.rec_cam[rix];
    const int slot = A.cam_slot[cam];
    const double2 xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    double d = 1.0, m = 0.0, a = 1.0;
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    RecLin L;
    linearize_record(A.camtab + (size_t)cam * kCamRec, X, psc, meta, xy.x, xy.y, d, m, a, A.loss, L);
    if (!L.ok) { bad += 1.0; continue; }
    double mrow[3] = {0, 0, 0};
    if (slot >= 0) {
      const double* y = A.yc + (size_t)slot * 6;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        mrow[q] = L.Jc[6 * q] * y[0] + L.Jc[6 * q + 1] * y[1] + L.Jc[6 * q + 2] * y[2] + L.Jc[6 * q + 3] * y[3] +
                  L.Jc[6 * q + 4] * y[4] + L.Jc[6 * q + 5] * y[5];
    }
    if (psc[0] != 0.0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double j0 = L.Jp[3 * q], j1 = L.Jp[3 * q + 1], j2 = L.Jp[3 * q + 2];
        acc[0] += j0 * j0; acc[1] += j0 * j1; acc[2] += j0 * j2; acc[3] += j1 * j1; acc[4] += j1 * j2; acc[5] += j2 * j2;
        const double rr = L.r[q] + mrow[q];
        acc[6] += j0 * rr; acc[7] += j1 * rr; acc[8] += j2 * rr;
      }
    }
  }
  block_sum<9>(acc, s_buf);
  double step_sq = 0.0, xn_sq = 0.0;
  if (tid == 0) {
    double yp[3] = {0, 0, 0}, X2[3] = {X[0], X[1], X[2]};
    if (pvar) {
      double V[6] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]}, Vi[6];
      V[0] += fmin(fmax(V[0], A.min_diag), A.max_diag) / lm_radius;
      V[3] += fmin(fmax(V[3], A.min_diag), A.max_diag) / lm_radius;
      V[5] += fmin(fmax(V[5], A.min_diag), A.max_diag) / lm_radius;
      if (!spd3_inverse(V, Vi)) {
        bad += 1.0;
      } else {
        sym3_mul(Vi, -acc[6], -acc[7], -acc[8], yp);
        for (int k = 0; k < 3; ++k) {
          const double dl = psc[k] * yp[k];
          X2[k] = X[k] + dl; step_sq += dl * dl; xn_sq += X2[k] * X2[k];
        }
      }
    }
    for (int k = 0; k < 3; ++k) { s_b[k] = yp[k]; s_b[3 + k] = X2[k]; A.pts2[3 * pix + k] = X2[k]; }
  }
  __syncthreads();
  const double y0 = s_b[0], y1 = s_b[1], y2 = s_b[2];
  const double X2[3] = {s_b[3], s_b[4], s_b[5]};
  double mcc = 0.0, cand = 0.0;
  for (int r = tid; r < H.nrec; r += kThreads) {
    const int rix = H.rec0 + r;
    const uint32_t meta = A.rec_meta[rix];
    const int cam = A.rec_cam[rix];
    const int slot = A.cam_slot[cam];
    const double2 xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    double d = 1.0, m = 0.0, a = 1.0;
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    RecLin L;
    linearize_record(A.camtab + (size_t)cam * kCamRec, X, psc, meta, xy.x, xy.y, d, m, a, A.loss, L);
    if (!L.ok) continue;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double mm = L.Jp[3 * q] * y0 + L.Jp[3 * q + 1] * y1 + L.Jp[3 * q + 2] * y2;
      if (slot >= 0) {
        const double* y = A.yc + (size_t)slot * 6;
        mm += L.Jc[6 * q] * y[0] + L.Jc[6 * q + 1] * y[1] + L.Jc[6 * q + 2] * y[2] + L.Jc[6 * q + 3] * y[3] +
              L.Jc[6 * q + 4] * y[4] + L.Jc[6 * q + 5] * y[5];
      }
      mcc -= mm * (L.r[q] + 0.5 * mm);
    }
    bool ok2 = true;
    const double cc = record_cost(A.camtab2 + (size_t)cam * kCamRec, X2, meta, xy.x, xy.y, d, m, a, A.loss, ok2);
    if (ok2) cand += cc; else bad += 1.0;
  }
  double red[5] = {cand, bad, mcc, step_sq, xn_sq};
  block_sum<5>(red, s_buf);
  if (tid < 5) A.part2[(size_t)(A.nchunks + blockIdx.x) * 8 + tid] = red[tid];
}

// cost of a record list at given cameras / landmarks (fixed blocks, eval_cost): per-block partials
__global__ __launch_bounds__(kThreads) void k_cost_records(CostArgs A) {
  __shared__ double s_red[3 * (kThreads / 64)];
  double cr = 0.0, cd = 0.0, bad = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < A.nrec; i += (int64_t)gridDim.x * kThreads) {
    const uint32_t meta = A.rec_meta[i];
    const int pt = A.rec_pt[i];
    const double X[3] = {A.pts[3 * pt], A.pts[3 * pt + 1], A.pts[3 * pt + 2]};
    const double* cam = A.camtab + (size_t)A.rec_cam[i] * kCamRec;
    bool ok = true;
    double d = 1.0, m = 0.0, a = 1.0;
    if (meta & kRecHasDepth) { d = A.rec_d[i]; m = A.rec_m[i]; a = A.rec_a[i]; }
    cr += record_cost(cam, X, meta & ~kRecHasDepth, A.rec_xy[2 * i], A.rec_xy[2 * i + 1], d, m, a, A.loss, ok);
    cd += record_cost(cam, X, meta & ~kRecHasReproj, 0.0, 0.0, d, m, a, A.loss, ok);
    if (!ok) bad += 1.0;
  }
  cr = wave_sum(cr); cd = wave_sum(cd); bad = wave_sum(bad);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[w] = cr; s_red[4 + w] = cd; s_red[8 + w] = bad; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double* s = &s_red[4 * threadIdx.x];
    A.part[(size_t)blockIdx.x * 4 + threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
  }
}

// deterministic column sums (or max for columns flagged in max_mask) of a [rows][stride] array
constexpr int kReduceThreads = 1024;
// fixed summation order (thread-strided rows, wave tree, then the 16 wave results in order): deterministic run to run.
// out2 (optional) receives a second copy.  s: 8 * kReduceThreads / 64 doubles of LDS, free again on return.
__device__ __forceinline__ void reduce_cols_block(const double* part, int64_t rows, int stride, int ncols, uint32_t max_mask, double* out, double* out2,
                                                  double* s) {
  constexpr int kWaves = kReduceThreads / 64;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t r = threadIdx.x; r < rows; r += kReduceThreads) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < ncols) {
        const double x = part[r * stride + c];
        v[c] = ((max_mask >> c) & 1u) ? fmax(v[c], x) : v[c] + x;
      }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < ncols) {
      const double w = ((max_mask >> c) & 1u) ? wave_max(v[c]) : wave_sum(v[c]);
      if ((threadIdx.x & 63) == 0) s[c * kWaves + (threadIdx.x >> 6)] = w;
    }
  __syncthreads();
  if ((int)threadIdx.x < ncols) {
    const double* p = &s[threadIdx.x * kWaves];
    const bool is_max = (max_mask >> threadIdx.x) & 1u;
    double r = p[0];
    for (int w = 1; w < kWaves; ++w) r = is_max ? fmax(r, p[w]) : r + p[w];
    out[threadIdx.x] = r;
    if (out2) out2[threadIdx.x] = r;
  }
  __syncthreads();
}
// gmax_slot >= 0 (sharded runs, the track sweep's scalars): the landmark-gradient maximum moves into this rank's slot of the
// tail behind the sums (k_gmax_to_slot's job: a MAX inside a SUM all-reduce)
__global__ __launch_bounds__(kReduceThreads) void k_reduce_cols(const double* part, int64_t rows, int stride, int ncols,
                                                                uint32_t max_mask, double* out, double* out2, int gmax_slot) {
  __shared__ double s[8 * (kReduceThreads / 64)];
  reduce_cols_block(part, rows, stride, ncols, max_mask, out, out2, s);
  if (gmax_slot >= 0 && threadIdx.x == 0) { out[SC_RANK0 + gmax_slot] = out[SC_GMAX_PTS]; out[SC_GMAX_PTS] = 0.0; }
}

// ---- camera-side kernels ------------------------------------------------------------------------
// camera table rows from (q, t, intrinsics, cs)
__global__ void k_build_camtab(int nc, const double* q, const double* t, const double* intr, const int32_t* intr_idx,
                               const double* cs, double* camtab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc) return;
  double* o = camtab + (size_t)i * kCamRec;
  quat_to_R(q + 4 * i, o);
  o[9] = t[3 * i]; o[10] = t[3 * i + 1]; o[11] = t[3 * i + 2];
  const double* K = intr + 4 * intr_idx[i];
  o[12] = K[0]; o[13] = K[1]; o[14] = K[2]; o[15] = K[3];
  for (int k = 0; k < 6; ++k) o[16 + k] = cs[6 * i + k];
  o[22] = o[23] = 0.0;
}

// Jacobi column scales: cs = mask / (1 + sqrt(diagU)), ps = 1 / (1 + sqrt(diagV)) (0 for constants)
__global__ void k_cam_scales(int nc, const int32_t* cam_slot, const double* cmask, const double* diagU, int jacobi,
                             double* cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc * 6) return;
  const int slot = cam_slot[i / 6];
  double v = 0.0;
  if (slot >= 0) v = jacobi ? cmask[i] / (1.0 + sqrt(diagU[(size_t)slot * 6 + (i % 6)])) : cmask[i];
  cs[i] = v;
}
__global__ void k_pt_scales(int64_t n3, const uint16_t* pt_kv, const double* diagV, int jacobi, double* ps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  double v = 0.0;
  if (pt_kv[i / 3] != 0xffff) v = jacobi ? 1.0 / (1.0 + sqrt(diagV[i])) : 1.0;
  ps[i] = v;
}

// candidate cameras q2,t2 = x [+] cs.*yc; ambient step / norm partials; camera gradient max norm.
// Single workgroup (cameras are few); writes scal[U_STEP_SQ_CAMS], [U_XN_SQ_CAMS], [U_GMAX_CAMS].
__global__ __launch_bounds__(kThreads) void k_cam_update(int nc, const int32_t* cam_slot, const double* q,
                                                         const double* t, const double* cs, const double* yc,
                                                         const double* gc, double* q2, double* t2, double* scal,
                                                         const double* intr, const int32_t* intr_idx, double* camtab2, int* chol_fail,
                                                         const LmCtl* ctl) {
  __shared__ double s_red[3 * (kThreads / 64)];
  if (lm_over(ctl)) return;
  CamUpdArgs U{0, nc, cam_slot, nullptr, q, t, cs, gc, intr, intr_idx, q2, t2, camtab2, scal, chol_fail};
  cam_update_all(U, yc, s_red);
}

// squared ambient norm of the variable landmarks (initial x_norm)
__global__ __launch_bounds__(kThreads) void k_pts_sqnorm(int64_t np, const uint16_t* pt_kv, const double* pts,
                                                         double* part) {
  __shared__ double s_red[kThreads / 64];
  double v = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < np; i += (int64_t)gridDim.x * kThreads)
    if (pt_kv[i] != 0xffff) v += pts[3 * i] * pts[3 * i] + pts[3 * i + 1] * pts[3 * i + 1] + pts[3 * i + 2] * pts[3 * i + 2];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Landmark-sharded runs: the tail of the reduced buffer is SUM-all-reduced, but the landmark gradient max-norm needs a MAX.
// Before the exchange every rank moves its value into its own slot (the others stay zero); afterwards the maximum over
// the slots is the global value.  Beyond kMaxRankSlots ranks the slot is shared (sum of maxima: conservative).
__global__ void k_gmax_to_slot(double* redsc, int rank) {
  if (threadIdx.x == 0) { redsc[SC_RANK0 + (rank % kMaxRankSlots)] = redsc[SC_GMAX_PTS]; redsc[SC_GMAX_PTS] = 0.0; }
}
__global__ void k_gmax_from_slots(const double* redsc, double* scal) {
  if (threadIdx.x == 0) {
    double m = 0.0;
    for (int r = 0; r < kMaxRankSlots; ++r) m = fmax(m, redsc[SC_RANK0 + r]);
    scal[U_X_COST] = redsc[SC_COST]; scal[U_X_BAD] = redsc[SC_BAD]; scal[U_GMAX_PTS] = m;
  }
}
void launch_gmax_to_slot(double* redsc, int rank, hipStream_t s) { hipLaunchKernelGGL(k_gmax_to_slot, dim3(1), dim3(64), 0, s, redsc, rank); }
void launch_gmax_from_slots(const double* redsc, double* scal, hipStream_t s) { hipLaunchKernelGGL(k_gmax_from_slots, dim3(1), dim3(64), 0, s, redsc, scal); }


// ---- Levenberg-Marquardt control on the device ------------------------------------------------------------------
// One thread takes the decisions of Ceres' TrustRegionMinimizer (trust_region_minimizer.cc, in its order) from the scalars
// of the iteration that just ran: evaluation of the step, function / parameter / gradient tolerance, acceptance, the
// radius rule of LevenbergMarquardtStrategy, the bookkeeping the summary reports.  The reference reaches this loop through
// pyceres.solve (mpsfm/sfm/mapper/bundle_adjustment.py:285-293) with Ceres' default options.
__device__ __forceinline__ void lm_decide_thread(LmCtl* C, const double* sc, const LmOpts& o);
// The host's copy of the control block, written straight into its pinned slot (device-visible host memory).  While the
// solve runs only the head travels (the host looks at the termination code and the diagnostics); the traces follow with the
// decision that ends the solve.
__device__ __forceinline__ void lm_copy_to_host(const LmCtl* C, LmCtl* host_copy) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(C);
  uint32_t* dst = reinterpret_cast<uint32_t*>(host_copy);
  const bool over = __hip_atomic_load(&C->term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kLmRunning;
  const int n = (int)((over ? sizeof(LmCtl) : sizeof(LmHead)) / 4);
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // no fence: the end of the kernel releases at system scope, and the host only looks after the event behind it
}
// redsc != NULL (sharded runs): the all-reduced scalars of the track sweep enter first — cost and invalid count summed, the
// landmark-gradient maximum over the rank slots (k_gmax_from_slots' job)
__global__ __launch_bounds__(64) void k_lm_decide(LmCtl* C, double* sc, LmOpts o, LmCtl* host_copy, const double* redsc) {
  const bool live = C->term == kLmRunning;
  if (live && threadIdx.x == 0 && redsc) {
    double m = 0.0;
    for (int r = 0; r < kMaxRankSlots; ++r) m = fmax(m, redsc[SC_RANK0 + r]);
    sc[U_X_COST] = redsc[SC_COST]; sc[U_X_BAD] = redsc[SC_BAD]; sc[U_GMAX_PTS] = m;
    __threadfence();  // lm_decide_thread reads the scalars past the L1
  }
  if (live && threadIdx.x == 0) lm_decide_thread(C, sc, o);
  __syncthreads();  // one workgroup: the barrier's workgroup-scope release / acquire orders thread 0's stores before the copy
  // the host's copy of the block goes straight into its pinned slot (device-visible host memory): no copy command in the loop
  if (host_copy) lm_copy_to_host(C, host_copy);
}
__device__ __forceinline__ void lm_decide_thread(LmCtl* C, const double* scal, const LmOpts& o) {
  double sc[U_COUNT];
#pragma unroll
  for (int i = 0; i < U_COUNT; ++i) sc[i] = __hip_atomic_load(scal + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // past this CU's L1
  LmHead L = *static_cast<const LmHead*>(C);
  lm_decide_logic(L, C, sc, o);
  *static_cast<LmHead*>(C) = L;
}

// Single-rank runs: the two partial-sum reductions of an iteration (track sweep: cost, bad count, landmark-gradient maximum;
// update sweep: candidate cost, bad count, model cost change, step and state norms) and its decision in ONE
// single-workgroup launch instead of three.
__global__ __launch_bounds__(kReduceThreads) void k_lm_reduce_decide(const double* part, const double* part2, int64_t rows, LmCtl* C, double* scal,
                                                                     LmOpts o, LmCtl* host_copy) {
  constexpr int kWaves = kReduceThreads / 64;
  __shared__ double s[8 * kWaves];
  __shared__ double s_out[8];
  const bool live = C->term == kLmRunning;
  // the scalars other kernels left (k_cam_update, the factorisation's failure flag): requested before the reductions
  double sc[U_COUNT];
  if (live && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < U_COUNT; ++i) sc[i] = __hip_atomic_load(scal + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // past this CU's L1
  }
  if (live && rows > 0) {
    // both partial tables in one pass, eight columns: 0-2 track sweep (cost, bad count, gradient maximum), 3-7 update sweep.
    // Fixed summation order (thread-strided rows, wave tree, then the wave results in order): deterministic run to run.
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = threadIdx.x; r < rows; r += kReduceThreads) {
      const double a0 = part[r * 4], a1 = part[r * 4 + 1], a2 = part[r * 4 + 2];
      const double b0 = part2[r * 8], b1 = part2[r * 8 + 1], b2 = part2[r * 8 + 2], b3 = part2[r * 8 + 3], b4 = part2[r * 8 + 4];
      v[0] += a0; v[1] += a1; v[2] = fmax(v[2], a2);
      v[3] += b0; v[4] += b1; v[5] += b2; v[6] += b3; v[7] += b4;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const double w = c == 2 ? wave_max(v[c]) : wave_sum(v[c]);
      if ((threadIdx.x & 63) == 0) s[c * kWaves + (threadIdx.x >> 6)] = w;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const double* p = &s[threadIdx.x * kWaves];
      double r = p[0];
      for (int w = 1; w < kWaves; ++w) r = threadIdx.x == 2 ? fmax(r, p[w]) : r + p[w];
      s_out[threadIdx.x] = r;
      scal[threadIdx.x < 3 ? U_X_COST + threadIdx.x : threadIdx.x - 3] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) sc[c < 3 ? U_X_COST + c : c - 3] = s_out[c];
    }
  }
  if (live && threadIdx.x == 0) {
    LmHead L = *static_cast<const LmHead*>(C);
    lm_decide_logic(L, C, sc, o);
    *static_cast<LmHead*>(C) = L;
  }
  // one workgroup: the barrier (workgroup-scope release / acquire) orders the deciding thread's stores before the copy;
  // agent-scope fences here cost an L2 write-back each (2-6 us)
  __syncthreads();
  if (host_copy) lm_copy_to_host(C, host_copy);
}

// First launch of an iteration: the candidate the previous iteration accepted becomes the state (copies instead of the
// pointer swaps a host-side loop would do; also in the iteration queued behind the last one, whose other kernels return at
// once) and the reduced buffer starts from zero.
__global__ __launch_bounds__(256) void k_lm_prologue(const LmCtl* C, double* red, int64_t nred, int nc, int64_t np, double* q, double* t, double* camtab,
                                                     double* pts, const double* q2, const double* t2, const double* camtab2, const double* pts2) {
  const bool acc = C->accepted != 0, live = C->term == kLmRunning;
  const int64_t n_pts = 3 * np, n_q = 4 * (int64_t)nc, n_t = 3 * (int64_t)nc, n_tab = (int64_t)kCamRec * nc;
  const int64_t total = acc ? n_pts + n_q + n_t + n_tab : 0;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
  for (int64_t i = i0; i < total; i += step) {
    if (i < n_pts) pts[i] = pts2[i];
    else if (i < n_pts + n_q) q[i - n_pts] = q2[i - n_pts];
    else if (i < n_pts + n_q + n_t) t[i - n_pts - n_q] = t2[i - n_pts - n_q];
    else camtab[i - n_pts - n_q - n_t] = camtab2[i - n_pts - n_q - n_t];
  }
  if (live)
    for (int64_t i = i0; i < nred; i += step) red[i] = 0.0;
}

// the accepted candidate becomes the state: after the loop (the iteration that ended the solve may have been accepted)
__global__ __launch_bounds__(256) void k_lm_accept(const LmCtl* C, int nc, int64_t np, double* q, double* t, double* camtab, double* pts,
                                                   const double* q2, const double* t2, const double* camtab2, const double* pts2) {
  if (!C->accepted) return;  // also after the last iteration: the step that met a tolerance is not taken
  const int64_t n_pts = 3 * np, n_q = 4 * (int64_t)nc, n_t = 3 * (int64_t)nc, n_tab = (int64_t)kCamRec * nc;
  const int64_t total = n_pts + n_q + n_t + n_tab;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    if (i < n_pts) pts[i] = pts2[i];
    else if (i < n_pts + n_q) q[i - n_pts] = q2[i - n_pts];
    else if (i < n_pts + n_q + n_t) t[i - n_pts - n_q] = t2[i - n_pts - n_q];
    else camtab[i - n_pts - n_q - n_t] = camtab2[i - n_pts - n_q - n_t];
  }
}

// landmark state between the caller's order and the handle's (chunk) order: dst[k] = src[perm[k]] / dst[perm[k]] = src[k]
__global__ __launch_bounds__(256) void k_permute_pts(int64_t np, const int32_t* perm, const double* src, double* dst, int scatter) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * np) return;
  const int64_t k = i / 3, c = i - 3 * k;
  const int64_t u = 3 * (int64_t)perm[k] + c;
  if (scatter) dst[u] = src[i];
  else dst[i] = src[u];
}
void launch_permute_pts(int64_t np, const int32_t* perm, const double* src, double* dst, bool scatter, hipStream_t s) {
  if (np <= 0) return;
  hipLaunchKernelGGL(k_permute_pts, dim3((unsigned)((3 * np + 255) / 256)), dim3(256), 0, s, np, perm, src, dst, scatter ? 1 : 0);
}

// The state norm and the cost of the fixed blocks enter the control block on the device (no host round trip in front of the
// loop): scal[U_XN_SQ_*] from k_cam_update / k_pts_sqnorm, scal[12..13] from the fixed blocks' cost reduction.  Sharded runs:
// k_lm_pack puts the three values that are sums over the ranks' landmarks into one small buffer, the exchange sums it, k_lm_init
// reads it back (the camera part of the norm is the same on every rank and stays out).
__global__ void k_lm_pack(const double* scal, double* sums) {
  if (threadIdx.x == 0) { sums[0] = scal[12]; sums[1] = scal[13]; sums[2] = scal[U_XN_SQ_PTS]; }
}
__global__ void k_lm_init(LmCtl* C, const double* scal, const double* sums) {
  if (threadIdx.x == 0) {
    const double f0 = sums ? sums[0] : scal[12], f1 = sums ? sums[1] : scal[13], xp = sums ? sums[2] : scal[U_XN_SQ_PTS];
    C->x_norm = sqrt(xp + scal[U_XN_SQ_CAMS]);
    C->fixed_cost = f0 + f1;
  }
}
void launch_lm_pack(const double* scal, double* sums, hipStream_t s) { hipLaunchKernelGGL(k_lm_pack, dim3(1), dim3(64), 0, s, scal, sums); }
void launch_lm_init(LmCtl* ctl, const double* scal, const double* sums, hipStream_t s) { hipLaunchKernelGGL(k_lm_init, dim3(1), dim3(64), 0, s, ctl, scal, sums); }

// ---- launch wrappers ------------------------------------------------------------------------------
void init_tile_tables(hipStream_t) {}

void launch_track_sweep(const SweepArgs& a, int nchunks, bool diag_only, hipStream_t s) {
  if (nchunks > 0) {
    if (diag_only) hipLaunchKernelGGL(k_track_sweep<MODE_DIAG>, dim3(nchunks), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(k_track_sweep<MODE_FULL>, dim3(nchunks), dim3(kThreads), 0, s, a);
  }
  if (a.nlong > 0) {
    if (diag_only) hipLaunchKernelGGL(k_long_track_sweep<MODE_DIAG>, dim3(a.nlong), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(k_long_track_sweep<MODE_FULL>, dim3(a.nlong), dim3(kThreads), 0, s, a);
  }
}
void launch_update_sweep(const SweepArgs& a, int nchunks, hipStream_t s, const CamUpdArgs* cu) {
  CamUpdArgs u{};
  if (cu) u = *cu;
  if (nchunks > 0) hipLaunchKernelGGL(k_update_sweep, dim3(nchunks), dim3(kThreads), 0, s, a, u);
  if (a.nlong > 0) hipLaunchKernelGGL(k_long_update_sweep, dim3(a.nlong), dim3(kThreads), 0, s, a);
}
void launch_cost_records(const CostArgs& a, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_cost_records, dim3(nblocks), dim3(kThreads), 0, s, a);
}
void launch_reduce_cols(const double* part, int64_t rows, int stride, int ncols, uint32_t max_mask, double* out,
                        hipStream_t s, double* out2, int gmax_slot) {
  hipLaunchKernelGGL(k_reduce_cols, dim3(1), dim3(kReduceThreads), 0, s, part, rows, stride, ncols, max_mask, out, out2, gmax_slot);
}
void launch_build_camtab(int nc, const double* q, const double* t, const double* intr, const int32_t* intr_idx,
                         const double* cs, double* camtab, hipStream_t s) {
  if (nc <= 0) return;
  hipLaunchKernelGGL(k_build_camtab, dim3((nc + 127) / 128), dim3(128), 0, s, nc, q, t, intr, intr_idx, cs, camtab);
}
void launch_cam_scales(int nc, const int32_t* cam_slot, const double* cmask, const double* diagU, int jacobi,
                       double* cs, hipStream_t s) {
  if (nc <= 0) return;
  hipLaunchKernelGGL(k_cam_scales, dim3((nc * 6 + 127) / 128), dim3(128), 0, s, nc, cam_slot, cmask, diagU, jacobi, cs);
}
void launch_pt_scales(int64_t np, const uint16_t* pt_kv, const double* diagV, int jacobi, double* ps, hipStream_t s) {
  if (np <= 0) return;
  const int64_t n3 = np * 3;
  hipLaunchKernelGGL(k_pt_scales, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, n3, pt_kv, diagV, jacobi, ps);
}
void launch_cam_update(int nc, const int32_t* cam_slot, const double* q, const double* t, const double* cs,
                       const double* yc, const double* gc, double* q2, double* t2, double* scal, hipStream_t s,
                       const double* intr, const int32_t* intr_idx, double* camtab2, int* chol_fail, const LmCtl* ctl) {
  hipLaunchKernelGGL(k_cam_update, dim3(1), dim3(kThreads), 0, s, nc, cam_slot, q, t, cs, yc, gc, q2, t2, scal, intr, intr_idx,
                     camtab2, chol_fail, ctl);
}
void launch_lm_decide(LmCtl* ctl, double* scal, const LmOpts& o, LmCtl* host_copy, hipStream_t s, const double* redsc) {
  hipLaunchKernelGGL(k_lm_decide, dim3(1), dim3(64), 0, s, ctl, scal, o, host_copy, redsc);
}
__global__ __launch_bounds__(256) void k_zero(double* p, int64_t n, const LmCtl* ctl) {
  if (lm_over(ctl)) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.0;
}
void launch_zero(double* p, int64_t n, const LmCtl* ctl, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_zero, dim3((unsigned)std::min<int64_t>(2048, (n + 255) / 256)), dim3(256), 0, s, p, n, ctl);
}
void launch_lm_reduce_decide(const double* part, const double* part2, int64_t rows, LmCtl* ctl, double* scal, const LmOpts& o, LmCtl* host_copy, hipStream_t s) {
  hipLaunchKernelGGL(k_lm_reduce_decide, dim3(1), dim3(kReduceThreads), 0, s, part, part2, rows, ctl, scal, o, host_copy);
}
void launch_lm_prologue(const LmCtl* ctl, double* red, int64_t nred, int nc, int64_t np, double* q, double* t, double* camtab, double* pts, const double* q2,
                        const double* t2, const double* camtab2, const double* pts2, hipStream_t s) {
  const int64_t total = std::max<int64_t>(nred, 3 * np + (int64_t)nc * (4 + 3 + kCamRec));
  const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
  hipLaunchKernelGGL(k_lm_prologue, dim3(grid), dim3(256), 0, s, ctl, red, nred, nc, np, q, t, camtab, pts, q2, t2, camtab2, pts2);
}
void launch_lm_accept(const LmCtl* ctl, int nc, int64_t np, double* q, double* t, double* camtab, double* pts, const double* q2, const double* t2,
                      const double* camtab2, const double* pts2, hipStream_t s) {
  const int64_t total = 3 * np + (int64_t)nc * (4 + 3 + kCamRec);
  const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (total + 255) / 256));
  hipLaunchKernelGGL(k_lm_accept, dim3(grid), dim3(256), 0, s, ctl, nc, np, q, t, camtab, pts, q2, t2, camtab2, pts2);
}
void launch_pts_sqnorm(int64_t np, const uint16_t* pt_kv, const double* pts, double* part, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_pts_sqnorm, dim3(nblocks), dim3(kThreads), 0, s, np, pt_kv, pts, part);
}

}  // namespace mpsfm
