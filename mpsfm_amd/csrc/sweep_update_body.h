// Body of the update sweep (see ba_kernels.hip), shared by k_update_sweep and the single-launch solver of small problems
// (local_lm.hip).
#pragma once
#include "common.h"
#include "sweep_common.h"

namespace mpsfm {

constexpr int kUpdLmCopies = 3, kUpdVCopy = kPtsMax * 6 + 2, kUpdGCopy = kPtsMax * 3 + 2;
// kUpdLmCopies copies of the landmark accumulators, chosen by the record's camera: the records of a landmark are neighbouring
// lanes, and same-address LDS atomics serialise
struct UpdLds {
  double V[kUpdLmCopies * kUpdVCopy];
  double g[kUpdLmCopies * kUpdGCopy];   // g_p + W^T y_c, then y_p
  double x2[kPtsMax * 3];               // candidate landmark
  int32_t slot[kLocalCamsMax];
  double red[5 * (kThreads / 64)];
};

// Update sweep of chunk `cix` by one workgroup of kThreads threads; yc: the camera steps by slot (HBM, or LDS for kLocal)
template <bool kLocal>
__device__ __forceinline__ void update_sweep_chunk(const SweepArgs& A, int cix, double lm_radius, const double* l_tab, const double* l_tab2,
                                                   const double* yc, UpdLds& S) {
  constexpr int kLmCopies = kUpdLmCopies, kVCopy = kUpdVCopy, kGCopy = kUpdGCopy;
  const int tid = thread_index<kLocal>();
  const ChunkHdr H = A.chunks[cix];
  const int nrec = H.nrec, npt = H.npt, ncam = H.ncam;
  for (int i = tid; i < kLmCopies * kVCopy; i += kThreads) S.V[i] = 0.0;
  for (int i = tid; i < kLmCopies * kGCopy; i += kThreads) S.g[i] = 0.0;
  if (tid < ncam) S.slot[tid] = A.chunk_cams[H.cam0 + tid];
  __syncthreads();

  RecUpd L;
  double mrow[3] = {0, 0, 0};
  uint32_t meta = 0;
  int cam = 0, lpt = 0;
  double2 xy = {0, 0};
  double d = 1.0, m = 0.0, a = 1.0;
  bool ok = true;
  if (tid < nrec) {
    const int rix = H.rec0 + tid;
    meta = A.rec_meta[rix];
    cam = A.rec_cam[rix];
    const int lcam = meta & 0xff;
    const int copy = lcam % kLmCopies;
    lpt = (meta >> 8) & 0xff;
    xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    const int pix = H.pt0 + lpt;
    const double X[3] = {A.pts[3 * pix], A.pts[3 * pix + 1], A.pts[3 * pix + 2]};
    const double psc[3] = {A.ps[3 * pix], A.ps[3 * pix + 1], A.ps[3 * pix + 2]};
    linearize_update(camera_row<kLocal>(A.camtab, l_tab, S.slot, cam, lcam), X, psc, meta, xy.x, xy.y, d, m, a, A.loss,
                     lcam != (int)kLcamConst ? yc + (size_t)S.slot[lcam] * 6 : nullptr, L);
    ok = L.ok;
    if (L.ok) {
      mrow[0] = L.mrow[0]; mrow[1] = L.mrow[1]; mrow[2] = L.mrow[2];
      if (psc[0] != 0.0) {
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double j0 = L.Jp[3 * r], j1 = L.Jp[3 * r + 1], j2 = L.Jp[3 * r + 2];
          V[0] += j0 * j0; V[1] += j0 * j1; V[2] += j0 * j2; V[3] += j1 * j1; V[4] += j1 * j2; V[5] += j2 * j2;
          const double rr = L.r[r] + mrow[r];
          g[0] += j0 * rr; g[1] += j1 * rr; g[2] += j2 * rr;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&S.V[copy * kVCopy + lpt * 6 + k], V[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&S.g[copy * kGCopy + lpt * 3 + k], g[k]);
      }
    }
  }
  __syncthreads();

  double step_sq = 0.0, xn_sq = 0.0;
  if (tid < npt) {
    const int pix = H.pt0 + tid;
    const double X[3] = {A.pts[3 * pix], A.pts[3 * pix + 1], A.pts[3 * pix + 2]};
    double yp[3] = {0, 0, 0}, X2[3] = {X[0], X[1], X[2]};
    if (A.pt_kv[pix] != 0xffff) {
      double V[6], Vi[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        V[k] = S.V[tid * 6 + k];
#pragma unroll
        for (int q = 1; q < kLmCopies; ++q) V[k] += S.V[q * kVCopy + tid * 6 + k];
      }
      V[0] += fmin(fmax(V[0], A.min_diag), A.max_diag) / lm_radius;
      V[3] += fmin(fmax(V[3], A.min_diag), A.max_diag) / lm_radius;
      V[5] += fmin(fmax(V[5], A.min_diag), A.max_diag) / lm_radius;
      if (!spd3_inverse(V, Vi)) {
        ok = false;
      } else {
        double gs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gs[k] = S.g[tid * 3 + k];
#pragma unroll
          for (int q = 1; q < kLmCopies; ++q) gs[k] += S.g[q * kGCopy + tid * 3 + k];
        }
        sym3_mul(Vi, -gs[0], -gs[1], -gs[2], yp);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double dl = A.ps[3 * pix + k] * yp[k];
          X2[k] = X[k] + dl;
          step_sq += dl * dl;
          xn_sq += X2[k] * X2[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      S.g[tid * 3 + k] = yp[k];
      S.x2[tid * 3 + k] = X2[k];
      A.pts2[3 * pix + k] = X2[k];
    }
  }
  __syncthreads();

  double mcc = 0.0, cand = 0.0;
  if (tid < nrec && ok) {
    const double y0 = S.g[lpt * 3], y1 = S.g[lpt * 3 + 1], y2 = S.g[lpt * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double mm = mrow[r] + L.Jp[3 * r] * y0 + L.Jp[3 * r + 1] * y1 + L.Jp[3 * r + 2] * y2;
      mcc -= mm * (L.r[r] + 0.5 * mm);
    }
    const double X2[3] = {S.x2[lpt * 3], S.x2[lpt * 3 + 1], S.x2[lpt * 3 + 2]};
    bool ok2 = true;
    cand = record_cost(camera_row<kLocal>(kLocal ? A.camtab : A.camtab2, l_tab2, S.slot, cam, (int)(meta & 0xff)), X2,  // (constant cameras have no candidate row of their own)
                       meta, xy.x, xy.y, d, m, a, A.loss, ok2);
    if (!ok2) { ok = false; cand = 0.0; }
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
