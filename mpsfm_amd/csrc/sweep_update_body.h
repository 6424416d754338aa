// Body of the update sweep (see ba_kernels.hip), shared by k_update_sweep and the single-launch solver of small problems
// (local_lm.hip).
#pragma once
#include "common.h"
#include "sweep_common.h"

namespace mpsfm {

// LDS of one chunk's update sweep (12 KB)
struct UpdLds {
  double V[kPtsMax * 6];
  double g[kPtsMax * 3];   // g_p + W^T y_c
  int32_t slot[kLocalCamsMax];
  double red[5 * (kThreads / 64)];
  uint8_t lpt[kThreads];   // landmark of every record (the first record of a landmark writes its candidate)
  double cand[kDenseCams * 16];  // fused camera update: candidate rows (R, t, K) of the chunk's cameras by local index
};

// Update sweep of chunk `cix` by one workgroup of kThreads threads; yc: the camera steps by slot (HBM, or LDS for kLocal).
//   1  one thread per record: the linearisation again (cheaper than 240 B per record through HBM), V_p and g_p + W^T y_c per landmark:
//      summed along the lanes first (the records of a landmark are neighbouring lanes: DPP segmented scan, as in the track
//      sweep), one LDS add per (16-lane row, landmark) run
//   2  EVERY record solves its landmark's 3x3 system itself (the ~5 records of a landmark repeat ~80 operations) instead of one
//      thread per landmark between two barriers: y_p, the candidate landmark — written by the landmark's first record —, the
//      model cost change and the candidate cost follow in the same registers
template <bool kLocal>
__device__ __forceinline__ void update_sweep_chunk(const SweepArgs& A, int cix, double lm_radius, const double* l_tab, const double* l_tab2,
                                                   const double* yc, UpdLds& S, const CamUpdArgs& U, bool fuse) {
  const int tid = thread_index<kLocal>();
  const ChunkHdr H = A.chunks[cix];
  const int nrec = H.nrec, npt = H.npt, ncam = H.ncam;
  for (int i = tid; i < npt * 6; i += kThreads) S.V[i] = 0.0;
  for (int i = tid; i < npt * 3; i += kThreads) S.g[i] = 0.0;
  if (tid < ncam) {
    const int slot = A.chunk_cams[H.cam0 + tid];
    S.slot[tid] = slot;
    if constexpr (!kLocal) {
      if (fuse && tid < kDenseCams) {  // the candidate row of local camera `tid`: the arithmetic of cam_update_all, so that the cost below is
        const int i = U.cam_of_slot[slot];  // evaluated at exactly the pose workgroup 0 writes out
        const double qq[4] = {U.q[4 * i], U.q[4 * i + 1], U.q[4 * i + 2], U.q[4 * i + 3]};
        const double tt[3] = {U.t[3 * i], U.t[3 * i + 1], U.t[3 * i + 2]};
        double qn[4], tn[3], s0 = 0.0, s1 = 0.0, s2 = 0.0;
        camera_candidate(qq, tt, U.cs + 6 * i, yc + (size_t)slot * 6, nullptr, qn, tn, s0, s1, s2);
        double* o = &S.cand[tid * 16];
        quat_to_R(qn, o);
        o[9] = tn[0]; o[10] = tn[1]; o[11] = tn[2];
        const double* K = U.intr + 4 * U.intr_idx[i];
        o[12] = K[0]; o[13] = K[1]; o[14] = K[2]; o[15] = K[3];
      }
    }
  }
  __syncthreads();

  RecUpd L;
  uint32_t meta = 0;
  int cam = 0, lpt = 0;
  double2 xy = {0, 0};
  double d = 1.0, m = 0.0, a = 1.0;
  double X[3] = {0, 0, 0}, psc[3] = {0, 0, 0};
  bool ok = true, variable = false;
  double Vg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < nrec) {
    const int rix = H.rec0 + tid;
    meta = A.rec_meta[rix];
    cam = A.rec_cam[rix];
    const int lcam = meta & 0xff;
    lpt = (meta >> 8) & 0xff;
    xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    const int pix = H.pt0 + lpt;
    variable = A.pt_kv[pix] != 0xffff;
#pragma unroll
    for (int k = 0; k < 3; ++k) { X[k] = A.pts[3 * pix + k]; psc[k] = A.ps[3 * pix + k]; }
    linearize_update(camera_row<kLocal>(A.camtab, l_tab, S.slot, cam, lcam), X, psc, meta, xy.x, xy.y, d, m, a, A.loss,
                     lcam != (int)kLcamConst ? yc + (size_t)S.slot[lcam] * 6 : nullptr, L);
    ok = L.ok;
    if (L.ok && psc[0] != 0.0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double j0 = L.Jp[3 * r], j1 = L.Jp[3 * r + 1], j2 = L.Jp[3 * r + 2];
        Vg[0] += j0 * j0; Vg[1] += j0 * j1; Vg[2] += j0 * j2; Vg[3] += j1 * j1; Vg[4] += j1 * j2; Vg[5] += j2 * j2;
        const double rr = L.r[r] + L.mrow[r];
        Vg[6] += j0 * rr; Vg[7] += j1 * rr; Vg[8] += j2 * rr;
      }
    }
  }
  S.lpt[tid] = tid < nrec ? (uint8_t)lpt : (uint8_t)255;
  {  // all lanes take part: threads beyond the records carry zeros and landmark 0
    seg_step<1>(lpt, Vg); seg_step<2>(lpt, Vg); seg_step<4>(lpt, Vg); seg_step<8>(lpt, Vg);
    const int nlpt = __builtin_amdgcn_update_dpp(-1, lpt, 0x101, 0xf, 0xf, false);  // row_shl:1: the right neighbour's landmark
    if (nlpt != lpt && lpt < npt) {  // last lane of its run inside the row (lane 15 of a row sees -1)
#pragma unroll
      for (int k = 0; k < 6; ++k) if (Vg[k] != 0.0) atomicAdd(&S.V[lpt * 6 + k], Vg[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) if (Vg[6 + k] != 0.0) atomicAdd(&S.g[lpt * 3 + k], Vg[6 + k]);
    }
  }
  __syncthreads();

  double step_sq = 0.0, xn_sq = 0.0, mcc = 0.0, cand = 0.0;
  if (tid < nrec) {
    const int pix = H.pt0 + lpt;
    double yp[3] = {0, 0, 0}, X2[3] = {X[0], X[1], X[2]};
    const bool first = tid == 0 || S.lpt[tid - 1] != (uint8_t)lpt;
    if (variable) {
      double V[6], F[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) V[k] = S.V[lpt * 6 + k];
      V[0] += fmin(fmax(V[0], A.min_diag), A.max_diag) / lm_radius;
      V[3] += fmin(fmax(V[3], A.min_diag), A.max_diag) / lm_radius;
      V[5] += fmin(fmax(V[5], A.min_diag), A.max_diag) / lm_radius;
      if (!spd3_inv_factor(V, F)) {  // F = chol(V + D)^-1, the factor the track sweep forms for the same block
        ok = false;  // (every record of the landmark reports it: the count only has to be non-zero)
      } else {
        const double g0 = S.g[lpt * 3], g1 = S.g[lpt * 3 + 1], g2 = S.g[lpt * 3 + 2];
        const double v0 = F[0] * g0, v1 = F[1] * g0 + F[2] * g1, v2 = F[3] * g0 + F[4] * g1 + F[5] * g2;  // F g
        yp[0] = -(F[0] * v0 + F[1] * v1 + F[3] * v2); yp[1] = -(F[2] * v1 + F[4] * v2); yp[2] = -(F[5] * v2);  // -F^T F g
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double dl = psc[k] * yp[k];
          X2[k] = X[k] + dl;
          if (first) { step_sq += dl * dl; xn_sq += X2[k] * X2[k]; }
        }
      }
    }
    if (first) {
#pragma unroll
      for (int k = 0; k < 3; ++k) A.pts2[3 * pix + k] = X2[k];
    }
    if (ok) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double mm = L.mrow[r] + L.Jp[3 * r] * yp[0] + L.Jp[3 * r + 1] * yp[1] + L.Jp[3 * r + 2] * yp[2];
        mcc -= mm * (L.r[r] + 0.5 * mm);
      }
      bool ok2 = true;
      const int lcam = (int)(meta & 0xff);
      const double* row;
      if (!kLocal && fuse) row = lcam != (int)kLcamConst ? &S.cand[lcam * 16] : A.camtab + (size_t)cam * kCamRec;
      else row = camera_row<kLocal>(kLocal ? A.camtab : A.camtab2, l_tab2, S.slot, cam, lcam);  // (constant cameras have no candidate row of their own)
      cand = record_cost(row, X2, meta, xy.x, xy.y, d, m, a, A.loss, ok2);
      if (!ok2) { ok = false; cand = 0.0; }
    }
  }
  const double r0 = wave_sum(cand), r1 = wave_sum(ok ? 0.0 : 1.0), r2 = wave_sum(mcc), r3 = wave_sum(step_sq),
               r4 = wave_sum(xn_sq);
  const int w = tid >> 6;
  if ((tid & 63) == 0) { S.red[w] = r0; S.red[4 + w] = r1; S.red[8 + w] = r2; S.red[12 + w] = r3; S.red[16 + w] = r4; }
  __syncthreads();
  if (tid < 5) {
    const double* s = &S.red[4 * tid];
    A.part2[(size_t)cix * 8 + tid] = (s[0] + s[1]) + (s[2] + s[3]);
  }
}

}  // namespace mpsfm
