// Device functions shared by the sweep kernels (ba_kernels.hip, sweep_dense.hip): residual blocks of one merged record
// (reprojection functor of pycolmap's bundle adjuster + the fork's log-depth functor, reference call sites
// mpsfm/sfm/mapper/bundle_adjustment.py:85-104, 163-176) with analytic Jacobians, Ceres' loss corrector folded in.
#pragma once
#include "common.h"

namespace mpsfm {

struct RecLin {
  double Jc[18];  // 3 x 6 (rows 0,1 reprojection, row 2 log-depth), robustified + scaled
  double Jp[9];   // 3 x 3
  double r[3];
  double cost;
  bool ok;
};

// residual blocks of one merged record at camera `cam` (table row) and landmark X
__device__ __forceinline__ void linearize_record(const double* __restrict__ cam, const double* X,
                                                 const double* psc, uint32_t meta, double u, double v,
                                                 double d, double m, double a, const LossParams& L,
                                                 RecLin& o) {
  double c[24];
  const double2* c2 = reinterpret_cast<const double2*>(cam);
#pragma unroll
  for (int i = 0; i < 11; ++i) { const double2 t = c2[i]; c[2 * i] = t.x; c[2 * i + 1] = t.y; }
  const double* R = c; const double* t = c + 9; const double* K = c + 12; const double* cs = c + 16;
  const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double Xc = Y0 + t[0], Yc = Y1 + t[1], Zc = Y2 + t[2];
  const double iz = fast_rcp(Zc);
  o.cost = 0.0;
  o.ok = true;
#pragma unroll
  for (int i = 0; i < 18; ++i) o.Jc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.Jp[i] = 0.0;
  o.r[0] = o.r[1] = o.r[2] = 0.0;
  if (meta & kRecHasReproj) {
    const double r0 = K[0] * Xc * iz + K[2] - u;
    const double r1 = K[1] * Yc * iz + K[3] - v;
    double rho0, rho1;
    loss_eval(L.reproj_type, L.reproj_a, r0 * r0 + r1 * r1, rho0, rho1);
    o.cost += 0.5 * L.reproj_mag * rho0;
    o.ok = o.ok && isfinite(r0) && isfinite(r1);
    const double w = fast_sqrt(L.reproj_mag * rho1);
    const double a00 = w * K[0] * iz, a02 = -w * K[0] * Xc * iz * iz;
    const double a11 = w * K[1] * iz, a12 = -w * K[1] * Yc * iz * iz;
    o.r[0] = w * r0; o.r[1] = w * r1;
    o.Jc[0] = a02 * (2 * Y1) * cs[0];
    o.Jc[1] = (a00 * (2 * Y2) - a02 * (2 * Y0)) * cs[1];
    o.Jc[2] = -a00 * (2 * Y1) * cs[2];
    o.Jc[3] = a00 * cs[3];
    o.Jc[5] = a02 * cs[5];
    o.Jc[6] = (a12 * (2 * Y1) - a11 * (2 * Y2)) * cs[0];
    o.Jc[7] = -a12 * (2 * Y0) * cs[1];
    o.Jc[8] = a11 * (2 * Y0) * cs[2];
    o.Jc[10] = a11 * cs[4];
    o.Jc[11] = a12 * cs[5];
    o.Jp[0] = (a00 * R[0] + a02 * R[6]) * psc[0];
    o.Jp[1] = (a00 * R[1] + a02 * R[7]) * psc[1];
    o.Jp[2] = (a00 * R[2] + a02 * R[8]) * psc[2];
    o.Jp[3] = (a11 * R[3] + a12 * R[6]) * psc[0];
    o.Jp[4] = (a11 * R[4] + a12 * R[7]) * psc[1];
    o.Jp[5] = (a11 * R[5] + a12 * R[8]) * psc[2];
  }
  if (meta & kRecHasDepth) {
    if (!(Zc > 0.0)) {
      o.ok = false;
    } else {
      const double rd = log(Zc) - d;  // d: log of the prior depth (ba_solver.hip)
      double rho0, rho1;
      loss_eval(L.depth_type, a, rd * rd, rho0, rho1);
      o.cost += 0.5 * m * rho0;
      const double sw = fast_sqrt(m * rho1);
      const double w = sw * iz;
      o.r[2] = sw * rd;
      o.Jc[12] = w * (2 * Y1) * cs[0];
      o.Jc[13] = -w * (2 * Y0) * cs[1];
      o.Jc[17] = w * cs[5];
      o.Jp[6] = w * R[6] * psc[0];
      o.Jp[7] = w * R[7] * psc[1];
      o.Jp[8] = w * R[8] * psc[2];
    }
  }
}

// The update sweep's view of a record: rows of Jp, the robustified residuals and mrow = Jc yc, with the camera Jacobian
// folded into the products (ys = cs .* yc) instead of materialised — 36 registers less at the kernel's pressure peak, which
// is what held it at four waves per SIMD — and without the loss value (one logarithm less per Cauchy block).
struct RecUpd {
  double Jp[9], r[3], mrow[3];
  bool ok;
};
__device__ __forceinline__ void linearize_update(const double* __restrict__ cam, const double* X, const double* psc, uint32_t meta, double u, double v,
                                                 double d, double m, double a, const LossParams& L, const double* yc, RecUpd& o) {
  double c[24];
  const double2* c2 = reinterpret_cast<const double2*>(cam);
#pragma unroll
  for (int i = 0; i < 11; ++i) { const double2 t = c2[i]; c[2 * i] = t.x; c[2 * i + 1] = t.y; }
  const double* R = c; const double* t = c + 9; const double* K = c + 12; const double* cs = c + 16;
  const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double Xc = Y0 + t[0], Yc = Y1 + t[1], Zc = Y2 + t[2];
  const double iz = fast_rcp(Zc);
  double ys[6] = {0, 0, 0, 0, 0, 0};
  if (yc) {
#pragma unroll
    for (int k = 0; k < 6; ++k) ys[k] = cs[k] * yc[k];
  }
  o.ok = true;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.Jp[i] = 0.0;
  o.r[0] = o.r[1] = o.r[2] = 0.0;
  o.mrow[0] = o.mrow[1] = o.mrow[2] = 0.0;
  if (meta & kRecHasReproj) {
    const double r0 = K[0] * Xc * iz + K[2] - u;
    const double r1 = K[1] * Yc * iz + K[3] - v;
    double rho0, rho1;
    loss_eval(L.reproj_type, L.reproj_a, r0 * r0 + r1 * r1, rho0, rho1);
    o.ok = o.ok && isfinite(r0) && isfinite(r1);
    const double w = fast_sqrt(L.reproj_mag * rho1);
    const double a00 = w * K[0] * iz, a02 = -w * K[0] * Xc * iz * iz;
    const double a11 = w * K[1] * iz, a12 = -w * K[1] * Yc * iz * iz;
    o.r[0] = w * r0; o.r[1] = w * r1;
    // the rows of Jc as in linearize_record, times ys
    o.mrow[0] = a02 * (2 * Y1) * ys[0] + (a00 * (2 * Y2) - a02 * (2 * Y0)) * ys[1] + (-a00 * (2 * Y1)) * ys[2] + a00 * ys[3] + a02 * ys[5];
    o.mrow[1] = (a12 * (2 * Y1) - a11 * (2 * Y2)) * ys[0] + (-a12 * (2 * Y0)) * ys[1] + a11 * (2 * Y0) * ys[2] + a11 * ys[4] + a12 * ys[5];
    o.Jp[0] = (a00 * R[0] + a02 * R[6]) * psc[0];
    o.Jp[1] = (a00 * R[1] + a02 * R[7]) * psc[1];
    o.Jp[2] = (a00 * R[2] + a02 * R[8]) * psc[2];
    o.Jp[3] = (a11 * R[3] + a12 * R[6]) * psc[0];
    o.Jp[4] = (a11 * R[4] + a12 * R[7]) * psc[1];
    o.Jp[5] = (a11 * R[5] + a12 * R[8]) * psc[2];
  }
  if (meta & kRecHasDepth) {
    if (!(Zc > 0.0)) {
      o.ok = false;
    } else {
      const double rd = log(Zc) - d;  // d: log of the prior depth (ba_solver.hip)
      // the robust weight alone (loss_eval's rho1): the loss value is not needed here
      double rho1 = 1.0;
      if (L.depth_type == MPSFM_LOSS_SOFT_L1) { double t, it; fast_sqrt_rsqrt(1.0 + rd * rd * fast_rcp(a * a), t, it); rho1 = fmax(DBL_MIN, it); }
      else if (L.depth_type == MPSFM_LOSS_CAUCHY) rho1 = fmax(DBL_MIN, fast_rcp(1.0 + rd * rd * fast_rcp(a * a)));
      const double sw = fast_sqrt(m * rho1);
      const double w = sw * iz;
      o.r[2] = sw * rd;
      o.mrow[2] = w * (2 * Y1) * ys[0] + (-w * (2 * Y0)) * ys[1] + w * ys[5];
      o.Jp[6] = w * R[6] * psc[0];
      o.Jp[7] = w * R[7] * psc[1];
      o.Jp[8] = w * R[8] * psc[2];
    }
  }
}

// cost only (candidate point)
__device__ __forceinline__ double record_cost(const double* __restrict__ cam, const double* X, uint32_t meta,
                                              double u, double v, double d, double m, double a,
                                              const LossParams& L, bool& ok) {
  const double Xc = cam[0] * X[0] + cam[1] * X[1] + cam[2] * X[2] + cam[9];
  const double Yc = cam[3] * X[0] + cam[4] * X[1] + cam[5] * X[2] + cam[10];
  const double Zc = cam[6] * X[0] + cam[7] * X[1] + cam[8] * X[2] + cam[11];
  double cost = 0.0;
  if (meta & kRecHasReproj) {
    const double iz = fast_rcp(Zc);
    const double r0 = cam[12] * Xc * iz + cam[14] - u;
    const double r1 = cam[13] * Yc * iz + cam[15] - v;
    double rho0, rho1;
    loss_eval(L.reproj_type, L.reproj_a, r0 * r0 + r1 * r1, rho0, rho1);
    cost += 0.5 * L.reproj_mag * rho0;
    ok = ok && isfinite(r0) && isfinite(r1);
  }
  if (meta & kRecHasDepth) {
    if (!(Zc > 0.0)) {
      ok = false;
    } else {
      const double rd = log(Zc) - d;  // d: log of the prior depth (ba_solver.hip)
      double rho0, rho1;
      loss_eval(L.depth_type, a, rd * rd, rho0, rho1);
      cost += 0.5 * m * rho0;
    }
  }
  return cost;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// camera table row of a record: the table in HBM, or (kLocal: single-launch solver) the workgroup's own LDS copy of the variable
// cameras, indexed by slot
template <bool kLocal>
__device__ __forceinline__ const double* camera_row(const double* tab, const double* l_tab, const int32_t* slot, int cam, int lcam) {
  if constexpr (kLocal) { if (lcam != (int)kLcamConst) return l_tab + slot[lcam] * kCamRec; }
  return tab + (size_t)cam * kCamRec;
}

// One step of the segmented scan along the lanes of a 16-lane row: adds the values of the lane SH to the left (DPP row_shr) when
// it belongs to the same landmark.  Lanes without such a neighbour in the row receive landmark -1.
template <int SH>
__device__ __forceinline__ void seg_step(int my_lpt, double (&v)[9]) {
  constexpr int ctrl = 0x110 | SH;
  const bool same = __builtin_amdgcn_update_dpp(-1, my_lpt, ctrl, 0xf, 0xf, false) == my_lpt;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[k]), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[k]), ctrl, 0xf, 0xf, false);
    v[k] += same ? __hiloint2double(hi, lo) : 0.0;
  }
}

// Candidate pose of camera i: x [+] cs .* y (quaternion: Ceres' EigenQuaternionManifold plus), and the camera's share of the
// step norm, the state norm and the gradient maximum (k_cam_update, the fused update sweep and local_lm.hip use the same arithmetic)
__device__ __forceinline__ void camera_candidate(const double* qq, const double* tt, const double* cs6, const double* y6, const double* gc6, double (&qn)[4],
                                                 double (&tn)[3], double& step, double& xn, double& gmax) {
  double dl[6], g[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double s = cs6[k];
    dl[k] = s * y6[k];
    g[k] = (gc6 && s > 0.0) ? -gc6[k] / s : 0.0;
  }
  quat_plus(qq, dl, qn);
#pragma unroll
  for (int k = 0; k < 3; ++k) tn[k] = tt[k] + dl[3 + k];
  if (gc6) {
    double qg[4];
    quat_plus(qq, g, qg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double d = qn[k] - qq[k];
      step += d * d; xn += qn[k] * qn[k];
      gmax = fmax(gmax, fabs(qg[k] - qq[k]));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double d = tn[k] - tt[k];
      step += d * d; xn += tn[k] * tn[k];
      gmax = fmax(gmax, fabs(g[3 + k]));
    }
  }
}

// what k_cam_update does, by one workgroup of kThreads threads; s_red: 12 doubles of LDS
__device__ __forceinline__ void cam_update_all(const CamUpdArgs& U, const double* yc, double* s_red) {
  double step = 0.0, xn = 0.0, gmax = 0.0;
  for (int i = threadIdx.x; i < U.nc; i += kThreads) {
    const int slot = U.cam_slot[i];
    const double qq[4] = {U.q[4 * i], U.q[4 * i + 1], U.q[4 * i + 2], U.q[4 * i + 3]};
    const double tt[3] = {U.t[3 * i], U.t[3 * i + 1], U.t[3 * i + 2]};
    double qn[4] = {qq[0], qq[1], qq[2], qq[3]}, tn[3] = {tt[0], tt[1], tt[2]};
    if (slot >= 0) camera_candidate(qq, tt, U.cs + 6 * i, yc + (size_t)slot * 6, U.gc + (size_t)slot * 6, qn, tn, step, xn, gmax);
    for (int k = 0; k < 4; ++k) U.q2[4 * i + k] = qn[k];
    for (int k = 0; k < 3; ++k) U.t2[3 * i + k] = tn[k];
    if (U.camtab2) {  // candidate camera table row (what k_build_camtab would write)
      double* o = U.camtab2 + (size_t)i * kCamRec;
      quat_to_R(qn, o);
      o[9] = tn[0]; o[10] = tn[1]; o[11] = tn[2];
      const double* K = U.intr + 4 * U.intr_idx[i];
      o[12] = K[0]; o[13] = K[1]; o[14] = K[2]; o[15] = K[3];
      for (int k = 0; k < 6; ++k) o[16 + k] = U.cs[6 * i + k];
      o[22] = o[23] = 0.0;
    }
  }
  step = wave_sum(step); xn = wave_sum(xn); gmax = wave_max(gmax);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[w] = step; s_red[4 + w] = xn; s_red[8 + w] = gmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    U.scal[U_STEP_SQ_CAMS] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    U.scal[U_XN_SQ_CAMS] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    U.scal[U_GMAX_CAMS] = fmax(fmax(s_red[8], s_red[9]), fmax(s_red[10], s_red[11]));
    // the factorisation's failure flag travels to the host with the other scalars and is re-armed here
    if (U.chol_fail) { U.scal[U_CHOL_FAIL] = (double)*U.chol_fail; *U.chol_fail = 0; }
  }
  __syncthreads();
}

// threadIdx.x; kOpaque: behind an empty asm, so that inside a loop over LM iterations (local_lm.hip) nothing derived from it is
// hoisted out of the loop and kept alive across every phase
template <bool kOpaque>
__device__ __forceinline__ int thread_index() {
  int t = threadIdx.x;
  if constexpr (kOpaque) asm volatile("" : "+v"(t));
  return t;
}

enum { MODE_FULL = 0, MODE_DIAG = 1 };

}  // namespace mpsfm
