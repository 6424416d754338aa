/*
 * ba_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A double-precision CPU restatement (C99 + OpenMP) of the nonlinear least-squares path that
 * MP-SfM's Optimizer drives through pycolmap/pyceres:
 *   problem assembly semantics  reference mpsfm/sfm/mapper/bundle_adjustment.py:67-185
 *   solver call                  reference mpsfm/sfm/mapper/bundle_adjustment.py:285-293
 *   point covariances            reference mpsfm/sfm/mapper/bundle_adjustment.py:244-261
 *
 * The arithmetic itself lives in third-party code that is NOT in the reference tree and not
 * installed here: pyceres v2.4 (docker/install_pyceres.sh:3), Ceres Solver 2.1.0
 * (docker/install_ceres_solver.sh:12) and an unpinned COLMAP fork (docker/install_colmap.sh:28).
 * This file restates their published algorithm:
 *   - Ceres 2.1 TrustRegionMinimizer + LevenbergMarquardtStrategy with default options
 *     (radius 1e4, diagonal clamp [1e-6,1e32], Jacobi column scaling from iteration 0,
 *     step quality update, ftol 1e-6 / gtol 1e-10 / ptol 1e-8, <= 50 iterations, <= 5
 *     consecutive invalid steps), Corrector with rho'' <= 0 (residual and Jacobian scaled by
 *     sqrt(rho')), TrivialLoss / SoftLOneLoss / CauchyLoss / ScaledLoss,
 *     EigenQuaternionManifold (q [+] d = exp(d) * q, no 1/2 factor), SubsetManifold.
 *   - Schur elimination of the point blocks and an exact Cholesky solve of the reduced camera
 *     system (what SPARSE_SCHUR computes, dense here).
 *   - COLMAP ReprojErrorCostFunctor<PINHOLE>: r = [fx X/Z + cx - u, fy Y/Z + cy - v],
 *     [X Y Z] = R(q) P + t.
 *   - the fork-only depth functor, INFERRED from its call site (:163-176, logloss=True):
 *     r = log Z - log(d * exp(s) + b) with (b, s) constant.
 *
 * PARITY UNPINNED against the reference: the reference ships no tests, fixtures or golden
 * outputs for this path, and pyceres/pycolmap cannot be installed offline.  The oracle is
 * instead pinned against independent implementations of the same objective
 * (scipy.optimize.least_squares soft_l1/cauchy, torch.autograd float64 Jacobians) by
 * tests/golden/make_golden.py and tests/test_oracle_*.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/mpsfm_hip.h"

#define ORACLE_API __attribute__((visibility("default")))

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------ */
/* loss functions: Ceres loss_function.cc semantics, rho[0..1] only (rho'' is never used
 * because rho'' <= 0 for all three, see corrector.cc). */
static inline void loss_eval(int type, double a, double s, double* rho0, double* rho1) {
  if (type == MPSFM_LOSS_SOFT_L1) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double tmp = sqrt(sum);
    *rho0 = 2.0 * b * (tmp - 1.0);
    *rho1 = fmax(DBL_MIN, 1.0 / tmp);
  } else if (type == MPSFM_LOSS_CAUCHY) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    *rho0 = b * log(sum);
    *rho1 = fmax(DBL_MIN, inv);
  } else {
    *rho0 = s;
    *rho1 = 1.0;
  }
}

ORACLE_API void oracle_loss(int type, double a, double s, double* out2) {
  loss_eval(type, a, s, &out2[0], &out2[1]);
}

/* ------------------------------------------------------------------------------------ */
/* geometry */
static inline void quat_to_R(const double q[4], double R[9]) {
  /* Eigen::Quaternion::toRotationMatrix, q = (x,y,z,w) */
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* EigenQuaternionManifold::Plus: q' = exp(d) * q, exp(d) = [sin|d| d/|d|, cos|d|] */
static inline void quat_plus(const double q[4], const double d[3], double out[4]) {
  const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0.0) { out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3]; return; }
  const double sbd = sin(n) / n;
  const double px = sbd * d[0], py = sbd * d[1], pz = sbd * d[2], pw = cos(n);
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  /* Hamilton product p * q */
  out[0] = pw * qx + px * qw + py * qz - pz * qy;
  out[1] = pw * qy - px * qz + py * qw + pz * qx;
  out[2] = pw * qz + px * qy - py * qx + pz * qw;
  out[3] = pw * qw - px * qx - py * qy - pz * qz;
}

/* One residual block.  kind 0: reprojection (2 rows), kind 1: log-depth (1 row).
 * Jc: rows x 6 (d theta, d t) in the tangent space of (EigenQuaternionManifold, R^3);
 * Jp: rows x 3.  Returns 0 when the block cannot be evaluated (depth block with Z <= 0). */
static inline int block_eval(int kind, const double R[9], const double t[3], const double K[4],
                             const double X[3], const double* meas, double deff,
                             double* r, double* Jc, double* Jp) {
  const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double Xc = Y0 + t[0], Yc = Y1 + t[1], Zc = Y2 + t[2];
  if (kind == 0) {
    const double iz = 1.0 / Zc;
    r[0] = K[0] * Xc * iz + K[2] - meas[0];
    r[1] = K[1] * Yc * iz + K[3] - meas[1];
    if (Jc) {
      const double a00 = K[0] * iz, a02 = -K[0] * Xc * iz * iz;
      const double a11 = K[1] * iz, a12 = -K[1] * Yc * iz * iz;
      /* dXc/dtheta = -2 [Y]x = [[0, 2Y2, -2Y1], [-2Y2, 0, 2Y0], [2Y1, -2Y0, 0]] */
      Jc[0] = a02 * (2 * Y1);              Jc[1] = a00 * (2 * Y2) + a02 * (-2 * Y0); Jc[2] = a00 * (-2 * Y1);
      Jc[3] = a00;                          Jc[4] = 0.0;                              Jc[5] = a02;
      Jc[6] = a11 * (-2 * Y2) + a12 * (2 * Y1); Jc[7] = a12 * (-2 * Y0);              Jc[8] = a11 * (2 * Y0);
      Jc[9] = 0.0;                          Jc[10] = a11;                             Jc[11] = a12;
      Jp[0] = a00 * R[0] + a02 * R[6]; Jp[1] = a00 * R[1] + a02 * R[7]; Jp[2] = a00 * R[2] + a02 * R[8];
      Jp[3] = a11 * R[3] + a12 * R[6]; Jp[4] = a11 * R[4] + a12 * R[7]; Jp[5] = a11 * R[5] + a12 * R[8];
    }
    return isfinite(r[0]) && isfinite(r[1]);
  } else {
    if (!(Zc > 0.0) || !(deff > 0.0)) return 0;
    r[0] = log(Zc) - log(deff);
    if (Jc) {
      const double iz = 1.0 / Zc;
      Jc[0] = iz * (2 * Y1); Jc[1] = iz * (-2 * Y0); Jc[2] = 0.0;
      Jc[3] = 0.0;           Jc[4] = 0.0;            Jc[5] = iz;
      Jp[0] = iz * R[6]; Jp[1] = iz * R[7]; Jp[2] = iz * R[8];
    }
    return isfinite(r[0]);
  }
}

/* exported single-block evaluators for the autograd known-answer tests */
ORACLE_API int oracle_reproj_block(const double* q, const double* t, const double* K,
                                   const double* X, const double* xy, double* r, double* Jc,
                                   double* Jp) {
  double R[9];
  quat_to_R(q, R);
  return block_eval(0, R, t, K, X, xy, 0.0, r, Jc, Jp);
}
ORACLE_API int oracle_depth_block(const double* q, const double* t, const double* X, double d,
                                  double shift, double logscale, double* r, double* Jc,
                                  double* Jp) {
  double R[9];
  quat_to_R(q, R);
  return block_eval(1, R, t, NULL, X, NULL, d * exp(logscale) + shift, r, Jc, Jp);
}
ORACLE_API void oracle_quat_plus(const double* q, const double* d, double* out) {
  quat_plus(q, d, out);
}

/* 3x3 SPD inverse through an LL^T factorisation (Ceres InvertPSDMatrix for fixed-size blocks
 * uses Eigen's LLT).  V: packed upper [00 01 02 11 12 22].  Returns 0 if not PD. */
static inline int spd3_inverse(const double V[6], double Vi[6]) {
  const double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5];
  if (!(a > 0.0)) return 0;
  const double l00 = sqrt(a);
  const double l10 = b / l00, l20 = c / l00;
  const double t11 = d - l10 * l10;
  if (!(t11 > 0.0)) return 0;
  const double l11 = sqrt(t11);
  const double l21 = (e - l20 * l10) / l11;
  const double t22 = f - l20 * l20 - l21 * l21;
  if (!(t22 > 0.0)) return 0;
  const double l22 = sqrt(t22);
  /* inverse of L (lower) */
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  /* Vi = Li^T Li */
  Vi[0] = i00 * i00 + i10 * i10 + i20 * i20;
  Vi[1] = i10 * i11 + i20 * i21;
  Vi[2] = i20 * i22;
  Vi[3] = i11 * i11 + i21 * i21;
  Vi[4] = i21 * i22;
  Vi[5] = i22 * i22;
  return 1;
}
static inline void sym3_mul(const double S[6], const double v[3], double out[3]) {
  out[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
  out[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
  out[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}

/* ------------------------------------------------------------------------------------ */
/* dense blocked Cholesky (lower, row-major, in place) + solve; OpenMP on the trailing update */
#define CH_NB 48
static int chol_unblocked(double* A, int lda, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * lda + j];
    for (int k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
    if (!(d > 0.0) || !isfinite(d)) return 0;
    d = sqrt(d);
    A[j * lda + j] = d;
    const double id = 1.0 / d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * lda + j];
      for (int k = 0; k < j; ++k) s -= A[i * lda + k] * A[j * lda + k];
      A[i * lda + j] = s * id;
    }
  }
  return 1;
}
static int dense_cholesky(double* A, int n) {
  for (int j = 0; j < n; j += CH_NB) {
    const int jb = (n - j < CH_NB) ? n - j : CH_NB;
    if (!chol_unblocked(A + (size_t)j * n + j, n, jb)) return 0;
    const int rest = n - j - jb;
    if (rest <= 0) break;
    double* Ljj = A + (size_t)j * n + j;
    /* TRSM: rows below, X L^T = A  */
#pragma omp parallel for schedule(static)
    for (int i = j + jb; i < n; ++i) {
      double* row = A + (size_t)i * n + j;
      for (int c = 0; c < jb; ++c) {
        double s = row[c];
        for (int k = 0; k < c; ++k) s -= row[k] * Ljj[c * n + k];
        row[c] = s / Ljj[c * n + c];
      }
    }
    /* SYRK: trailing lower triangle -= X X^T */
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = j + jb; i < n; ++i) {
      const double* xi = A + (size_t)i * n + j;
      double* dst = A + (size_t)i * n;
      for (int c = j + jb; c <= i; ++c) {
        const double* xc = A + (size_t)c * n + j;
        double s = 0.0;
        for (int k = 0; k < jb; ++k) s += xi[k] * xc[k];
        dst[c] -= s;
      }
    }
  }
  return 1;
}
static void dense_chol_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    const double* row = L + (size_t)i * n;
    for (int k = 0; k < i; ++k) s -= row[k] * b[k];
    b[i] = s / row[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}

ORACLE_API int oracle_dense_spd_solve(double* A, int n, double* b) {
  if (!dense_cholesky(A, n)) return -1;
  dense_chol_solve(A, n, b);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
typedef struct {
  const mpsfm_ba_problem* P;
  int nc, np;
  int ncv, n;          /* variable cameras, reduced dimension 6*ncv */
  int* cam_sys;        /* [nc] system slot or -1 (constant / no residuals) */
  double* cmask;       /* [nc][6] tangent mask (0 for constant coordinates) */
  uint8_t* pvar;       /* [np] point is a variable of the reduced program */
  /* residual blocks grouped by point (CSR); fixed blocks (constant camera AND constant
   * point) are kept out and only add to fixed_cost */
  int64_t nblk;
  int64_t* pt_start;   /* [np+1] */
  int32_t* blk_cam;
  uint8_t* blk_kind;
  int64_t* blk_src;
  int64_t nfixed;
  int32_t* fx_cam; int32_t* fx_pt; uint8_t* fx_kind; int64_t* fx_src;
  /* per-solve work */
  double* cs;          /* [nc][6] Jacobi scale * mask */
  double* ps;          /* [np][3] */
  double* Jc;          /* [nblk][12] */
  double* Jp;          /* [nblk][6]  */
  double* rr;          /* [nblk][2]  */
  double* V;           /* [np][6]    */
  double* gp;          /* [np][3]    */
  /* all-reduced buffer: S (n*n) | gc (n) | diagU (n) | scalars(8) */
  double* red; int64_t red_count;
  double* S; double* gc; double* diagU; double* sc;
  double* rhs;         /* [n] */
  double* yc;          /* [n] */
  double* yp;          /* [np][3] */
  uint8_t* locks;      /* [ncv*ncv] cell spin locks */
  mpsfm_allreduce_fn allreduce; void* ar_user;
} ctx_t;

static int do_allreduce(ctx_t* c, double* buf, int64_t count) {
  if (!c->allreduce) return 0;
  return c->allreduce(c->ar_user, buf, count, 0, NULL);
}

static void ctx_free(ctx_t* c) {
  free(c->cam_sys); free(c->cmask); free(c->pvar); free(c->pt_start); free(c->blk_cam);
  free(c->blk_kind); free(c->blk_src); free(c->fx_cam); free(c->fx_pt); free(c->fx_kind);
  free(c->fx_src); free(c->cs); free(c->ps); free(c->Jc); free(c->Jp); free(c->rr); free(c->V);
  free(c->gp); free(c->red); free(c->rhs); free(c->yc); free(c->yp); free(c->locks);
  memset(c, 0, sizeof(*c));
}

static int check_problem(const mpsfm_ba_problem* P) {
  if (!P || P->n_cams < 0 || P->n_pts < 0 || P->n_obs < 0 || P->n_dobs < 0) return 0;
  for (int64_t i = 0; i < P->n_obs; ++i)
    if (P->obs_cam[i] < 0 || P->obs_cam[i] >= P->n_cams || P->obs_pt[i] < 0 || P->obs_pt[i] >= P->n_pts) return 0;
  for (int64_t i = 0; i < P->n_dobs; ++i)
    if (P->dobs_cam[i] < 0 || P->dobs_cam[i] >= P->n_cams || P->dobs_pt[i] < 0 || P->dobs_pt[i] >= P->n_pts) return 0;
  for (int i = 0; i < P->n_cams; ++i)
    if (P->cam_intr_idx[i] < 0 || P->cam_intr_idx[i] >= P->n_intr) return 0;
  return 1;
}

static int ctx_build(ctx_t* c, const mpsfm_ba_problem* P, mpsfm_allreduce_fn ar, void* aru) {
  memset(c, 0, sizeof(*c));
  c->P = P; c->nc = P->n_cams; c->np = P->n_pts; c->allreduce = ar; c->ar_user = aru;
  const int nc = c->nc, np = c->np;
  /* cameras with at least one residual block whose point is variable, or any block at all:
   * Ceres drops parameter blocks that appear in no residual block of the reduced program.
   * A camera is part of the reduced program iff it is not constant and has >= 1 block. */
  double* cnt = (double*)calloc((size_t)nc + 1, sizeof(double));
  for (int64_t i = 0; i < P->n_obs; ++i) cnt[P->obs_cam[i]] += 1.0;
  for (int64_t i = 0; i < P->n_dobs; ++i) cnt[P->dobs_cam[i]] += 1.0;
  if (do_allreduce(c, cnt, nc)) { free(cnt); return MPSFM_ECOMM; }
  c->cam_sys = (int*)malloc(sizeof(int) * (size_t)(nc + 1));
  c->cmask = (double*)calloc((size_t)nc * 6 + 6, sizeof(double));
  c->ncv = 0;
  for (int i = 0; i < nc; ++i) {
    if (P->pose_const[i] || cnt[i] == 0.0) { c->cam_sys[i] = -1; continue; }
    c->cam_sys[i] = c->ncv++;
    for (int k = 0; k < 6; ++k) c->cmask[i * 6 + k] = 1.0;
    if (i == P->gauge_axis_cam) c->cmask[i * 6 + 3] = 0.0;
  }
  free(cnt);
  c->n = 6 * c->ncv;
  /* group blocks by point */
  c->pt_start = (int64_t*)calloc((size_t)np + 2, sizeof(int64_t));
  c->pvar = (uint8_t*)calloc((size_t)np + 1, 1);
  int64_t nfix = 0;
  for (int64_t i = 0; i < P->n_obs; ++i) {
    const int cam = P->obs_cam[i], pt = P->obs_pt[i];
    if (c->cam_sys[cam] < 0 && P->pt_const[pt]) { ++nfix; continue; }
    c->pt_start[pt + 1]++;
  }
  for (int64_t i = 0; i < P->n_dobs; ++i) {
    const int cam = P->dobs_cam[i], pt = P->dobs_pt[i];
    if (c->cam_sys[cam] < 0 && P->pt_const[pt]) { ++nfix; continue; }
    c->pt_start[pt + 1]++;
  }
  for (int p = 0; p < np; ++p) {
    c->pvar[p] = (!P->pt_const[p] && c->pt_start[p + 1] > 0) ? 1 : 0;
    c->pt_start[p + 1] += c->pt_start[p];
  }
  c->nblk = c->pt_start[np];
  c->nfixed = nfix;
  c->blk_cam = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->nblk + 1));
  c->blk_kind = (uint8_t*)malloc((size_t)c->nblk + 1);
  c->blk_src = (int64_t*)malloc(sizeof(int64_t) * (size_t)(c->nblk + 1));
  c->fx_cam = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nfix + 1));
  c->fx_pt = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nfix + 1));
  c->fx_kind = (uint8_t*)malloc((size_t)nfix + 1);
  c->fx_src = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nfix + 1));
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(np + 1));
  memcpy(fill, c->pt_start, sizeof(int64_t) * (size_t)np);
  int64_t fi = 0;
  for (int64_t i = 0; i < P->n_obs; ++i) {
    const int cam = P->obs_cam[i], pt = P->obs_pt[i];
    if (c->cam_sys[cam] < 0 && P->pt_const[pt]) {
      c->fx_cam[fi] = cam; c->fx_pt[fi] = pt; c->fx_kind[fi] = 0; c->fx_src[fi] = i; ++fi; continue;
    }
    const int64_t k = fill[pt]++;
    c->blk_cam[k] = cam; c->blk_kind[k] = 0; c->blk_src[k] = i;
  }
  for (int64_t i = 0; i < P->n_dobs; ++i) {
    const int cam = P->dobs_cam[i], pt = P->dobs_pt[i];
    if (c->cam_sys[cam] < 0 && P->pt_const[pt]) {
      c->fx_cam[fi] = cam; c->fx_pt[fi] = pt; c->fx_kind[fi] = 1; c->fx_src[fi] = i; ++fi; continue;
    }
    const int64_t k = fill[pt]++;
    c->blk_cam[k] = cam; c->blk_kind[k] = 1; c->blk_src[k] = i;
  }
  free(fill);
  c->cs = (double*)calloc((size_t)nc * 6 + 6, sizeof(double));
  c->ps = (double*)calloc((size_t)np * 3 + 3, sizeof(double));
  c->Jc = (double*)malloc(sizeof(double) * 12 * (size_t)(c->nblk + 1));
  c->Jp = (double*)malloc(sizeof(double) * 6 * (size_t)(c->nblk + 1));
  c->rr = (double*)malloc(sizeof(double) * 2 * (size_t)(c->nblk + 1));
  c->V = (double*)malloc(sizeof(double) * 6 * (size_t)(np + 1));
  c->gp = (double*)malloc(sizeof(double) * 3 * (size_t)(np + 1));
  const size_t n = (size_t)c->n;
  c->red_count = (int64_t)(n * n + 2 * n + 8);
  c->red = (double*)malloc(sizeof(double) * (size_t)c->red_count);
  c->S = c->red; c->gc = c->red + n * n; c->diagU = c->gc + n; c->sc = c->diagU + n;
  c->rhs = (double*)malloc(sizeof(double) * (n + 1));
  c->yc = (double*)calloc(n + 1, sizeof(double));
  c->yp = (double*)calloc((size_t)np * 3 + 3, sizeof(double));
  c->locks = (uint8_t*)calloc((size_t)c->ncv * (size_t)c->ncv + 1, 1);
  if (!c->Jc || !c->Jp || !c->rr || !c->V || !c->gp || !c->red || !c->yp || !c->locks) return MPSFM_ENOMEM;
  return 0;
}

static inline double depth_eff(const mpsfm_ba_problem* P, int cam, int64_t src) {
  double b = 0.0, s = 0.0;
  if (P->shift_logscale) { b = P->shift_logscale[cam * 2]; s = P->shift_logscale[cam * 2 + 1]; }
  return P->dobs_depth[src] * exp(s) + b;
}

/* cost of one block (1/2 mag rho(|r|^2)); returns NaN when the block cannot be evaluated */
static inline double block_cost(const mpsfm_ba_problem* P, int kind, int cam, int pt, int64_t src,
                                const double* R, const double* t, const double* X) {
  double r[2];
  if (kind == 0) {
    if (!block_eval(0, R, t, P->cam_intr + 4 * P->cam_intr_idx[cam], X, P->obs_xy + 2 * src, 0.0, r, NULL, NULL))
      return NAN;
    double rho0, rho1;
    loss_eval(P->reproj_loss_type, P->reproj_loss_scale, r[0] * r[0] + r[1] * r[1], &rho0, &rho1);
    return 0.5 * P->reproj_loss_magnitude * rho0;
  } else {
    if (!block_eval(1, R, t, NULL, X, NULL, depth_eff(P, cam, src), r, NULL, NULL)) return NAN;
    double rho0, rho1;
    loss_eval(P->depth_loss_type, P->dobs_param[src], r[0] * r[0], &rho0, &rho1);
    return 0.5 * P->dobs_magnitude[src] * rho0;
  }
  (void)pt;
}

/* cost of the reduced program at (q, t, pts): local shard part, NaN if not evaluable.
 * cost_split (optional): [reproj, depth] */
static double eval_cost(ctx_t* c, const double* q, const double* t, const double* pts, double* split) {
  const mpsfm_ba_problem* P = c->P;
  double* Rall = (double*)malloc(sizeof(double) * 9 * (size_t)(c->nc + 1));
  for (int i = 0; i < c->nc; ++i) quat_to_R(q + 4 * i, Rall + 9 * i);
  double cr = 0.0, cd = 0.0;
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 512) reduction(+ : cr, cd) reduction(| : bad)
  for (int p = 0; p < c->np; ++p) {
    for (int64_t k = c->pt_start[p]; k < c->pt_start[p + 1]; ++k) {
      const int cam = c->blk_cam[k];
      const double v = block_cost(P, c->blk_kind[k], cam, p, c->blk_src[k], Rall + 9 * cam, t + 3 * cam, pts + 3 * p);
      if (!isfinite(v)) bad = 1;
      if (c->blk_kind[k] == 0) cr += v; else cd += v;
    }
  }
  free(Rall);
  if (split) { split[0] = cr; split[1] = cd; }
  if (bad) return NAN;
  return cr + cd;
}

static double eval_fixed_cost(ctx_t* c, const double* q, const double* t, const double* pts, double* split) {
  const mpsfm_ba_problem* P = c->P;
  double cr = 0.0, cd = 0.0;
  for (int64_t k = 0; k < c->nfixed; ++k) {
    const int cam = c->fx_cam[k], pt = c->fx_pt[k];
    double R[9];
    quat_to_R(q + 4 * cam, R);
    const double v = block_cost(P, c->fx_kind[k], cam, pt, c->fx_src[k], R, t + 3 * cam, pts + 3 * pt);
    if (c->fx_kind[k] == 0) cr += v; else cd += v;
  }
  if (split) { split[0] = cr; split[1] = cd; }
  return cr + cd;
}

/* Linearise at (q,t,pts) with column scales cs/ps: fills Jc/Jp/rr (robustified, scaled),
 * V, gp, and the all-reduced buffer S(diag blocks = U) | gc | diagU | sc[0] = cost.
 * Returns 0 if some block cannot be evaluated. */
static int linearize(ctx_t* c, const double* q, const double* t, const double* pts) {
  const mpsfm_ba_problem* P = c->P;
  const int n = c->n;
  memset(c->red, 0, sizeof(double) * (size_t)c->red_count);
  double* Rall = (double*)malloc(sizeof(double) * 9 * (size_t)(c->nc + 1));
  for (int i = 0; i < c->nc; ++i) quat_to_R(q + 4 * i, Rall + 9 * i);
  double cost = 0.0;
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 512) reduction(+ : cost) reduction(| : bad)
  for (int p = 0; p < c->np; ++p) {
    double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    const double* sp = c->ps + 3 * p;
    for (int64_t k = c->pt_start[p]; k < c->pt_start[p + 1]; ++k) {
      const int cam = c->blk_cam[k];
      const int kind = c->blk_kind[k];
      const int64_t src = c->blk_src[k];
      const int rows = kind == 0 ? 2 : 1;
      double r[2] = {0, 0}, Jc[12], Jp[6];
      int ok;
      double w, rho0, rho1;
      if (kind == 0) {
        ok = block_eval(0, Rall + 9 * cam, t + 3 * cam, P->cam_intr + 4 * P->cam_intr_idx[cam], pts + 3 * p,
                        P->obs_xy + 2 * src, 0.0, r, Jc, Jp);
        loss_eval(P->reproj_loss_type, P->reproj_loss_scale, r[0] * r[0] + r[1] * r[1], &rho0, &rho1);
        cost += 0.5 * P->reproj_loss_magnitude * rho0;
        w = sqrt(P->reproj_loss_magnitude * rho1);
      } else {
        ok = block_eval(1, Rall + 9 * cam, t + 3 * cam, NULL, pts + 3 * p, NULL, depth_eff(P, cam, src), r, Jc, Jp);
        loss_eval(P->depth_loss_type, P->dobs_param[src], r[0] * r[0], &rho0, &rho1);
        cost += 0.5 * P->dobs_magnitude[src] * rho0;
        w = sqrt(P->dobs_magnitude[src] * rho1);
      }
      if (!ok) { bad = 1; continue; }
      const double* sc = c->cs + 6 * cam;
      double* oJc = c->Jc + 12 * k;
      double* oJp = c->Jp + 6 * k;
      double* orr = c->rr + 2 * k;
      orr[0] = orr[1] = 0.0;
      for (int a = 0; a < rows; ++a) {
        orr[a] = w * r[a];
        for (int b = 0; b < 6; ++b) oJc[a * 6 + b] = w * Jc[a * 6 + b] * sc[b];
        for (int b = 0; b < 3; ++b) oJp[a * 3 + b] = c->pvar[p] ? w * Jp[a * 3 + b] * sp[b] : 0.0;
      }
      for (int a = rows; a < 2; ++a) {
        for (int b = 0; b < 6; ++b) oJc[a * 6 + b] = 0.0;
        for (int b = 0; b < 3; ++b) oJp[a * 3 + b] = 0.0;
      }
      /* point block */
      for (int a = 0; a < rows; ++a) {
        const double* jp = oJp + 3 * a;
        V[0] += jp[0] * jp[0]; V[1] += jp[0] * jp[1]; V[2] += jp[0] * jp[2];
        V[3] += jp[1] * jp[1]; V[4] += jp[1] * jp[2]; V[5] += jp[2] * jp[2];
        g[0] += jp[0] * orr[a]; g[1] += jp[1] * orr[a]; g[2] += jp[2] * orr[a];
      }
      /* camera block U_c, g_c */
      const int slot = c->cam_sys[cam];
      if (slot >= 0) {
        double U[36], gcl[6];
        for (int i = 0; i < 6; ++i) {
          gcl[i] = 0.0;
          for (int a = 0; a < rows; ++a) gcl[i] += oJc[a * 6 + i] * orr[a];
          for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int a = 0; a < rows; ++a) s += oJc[a * 6 + i] * oJc[a * 6 + j];
            U[i * 6 + j] = s;
          }
        }
        uint8_t* lk = &c->locks[(size_t)slot * c->ncv + slot];
        while (__atomic_test_and_set(lk, __ATOMIC_ACQUIRE)) { }
        for (int i = 0; i < 6; ++i) {
          double* row = c->S + (size_t)(6 * slot + i) * n + 6 * slot;
          for (int j = 0; j < 6; ++j) row[j] += U[i * 6 + j];
          c->gc[6 * slot + i] += gcl[i];
          c->diagU[6 * slot + i] += U[i * 6 + i];
        }
        __atomic_clear(lk, __ATOMIC_RELEASE);
      }
    }
    memcpy(c->V + 6 * p, V, sizeof(V));
    memcpy(c->gp + 3 * p, g, sizeof(g));
  }
  free(Rall);
  c->sc[0] = cost;
  return !bad;
}

/* Schur complement: S -= sum_p W_p (V_p + D_p)^-1 W_p^T ; rhs = -gc + sum_p W_p Vinv g_p.
 * Dp = clamp(diag V)/radius.  Local shard contribution; all-reduce happens in the caller.
 * Stores (V+D)^-1 back into c->V. */
static int schur_reduce(ctx_t* c, double radius, double min_d, double max_d) {
  const int n = c->n;
  int bad = 0;
  /* rhs part accumulates into gc with a sign flip later: use a separate buffer */
  double* rhs_add = (double*)calloc((size_t)n + 1, sizeof(double));
#pragma omp parallel
  {
    double* rloc = (double*)calloc((size_t)n + 1, sizeof(double));
    int cap = 64;
    int* cams = (int*)malloc(sizeof(int) * cap);
    double* W = (double*)malloc(sizeof(double) * 18 * cap);
#pragma omp for schedule(dynamic, 256)
    for (int p = 0; p < c->np; ++p) {
      if (!c->pvar[p]) continue;
      double* V = c->V + 6 * p;
      double Vd[6] = {V[0], V[1], V[2], V[3], V[4], V[5]}, Vi[6];
      Vd[0] += fmin(fmax(V[0], min_d), max_d) / radius;
      Vd[3] += fmin(fmax(V[3], min_d), max_d) / radius;
      Vd[5] += fmin(fmax(V[5], min_d), max_d) / radius;
      if (!spd3_inverse(Vd, Vi)) { bad = 1; continue; }
      memcpy(V, Vi, sizeof(Vi));
      /* gather W per distinct variable camera */
      const int64_t b0 = c->pt_start[p], b1 = c->pt_start[p + 1];
      int m = 0;
      for (int64_t k = b0; k < b1; ++k) {
        const int slot = c->cam_sys[c->blk_cam[k]];
        if (slot < 0) continue;
        int idx = -1;
        for (int i = 0; i < m; ++i) if (cams[i] == slot) { idx = i; break; }
        if (idx < 0) {
          if (m == cap) { cap *= 2; cams = (int*)realloc(cams, sizeof(int) * cap); W = (double*)realloc(W, sizeof(double) * 18 * cap); }
          idx = m++;
          cams[idx] = slot;
          for (int i = 0; i < 18; ++i) W[18 * idx + i] = 0.0;
        }
        const double* jc = c->Jc + 12 * k;
        const double* jp = c->Jp + 6 * k;
        for (int a = 0; a < 2; ++a)
          for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 3; ++j) W[18 * idx + 3 * i + j] += jc[a * 6 + i] * jp[a * 3 + j];
      }
      double Vg[3];
      sym3_mul(Vi, c->gp + 3 * p, Vg);
      for (int i = 0; i < m; ++i) {
        /* Y = W_i Vinv (6x3) */
        double Y[18];
        for (int a = 0; a < 6; ++a) sym3_mul(Vi, W + 18 * i + 3 * a, Y + 3 * a);
        for (int a = 0; a < 6; ++a)
          rloc[6 * cams[i] + a] += W[18 * i + 3 * a] * Vg[0] + W[18 * i + 3 * a + 1] * Vg[1] + W[18 * i + 3 * a + 2] * Vg[2];
        for (int j = 0; j < m; ++j) {
          if (cams[j] < cams[i]) continue; /* upper block triangle only */
          double B[36];
          for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b)
              B[a * 6 + b] = Y[3 * a] * W[18 * j + 3 * b] + Y[3 * a + 1] * W[18 * j + 3 * b + 1] + Y[3 * a + 2] * W[18 * j + 3 * b + 2];
          uint8_t* lk = &c->locks[(size_t)cams[i] * c->ncv + cams[j]];
          while (__atomic_test_and_set(lk, __ATOMIC_ACQUIRE)) { }
          for (int a = 0; a < 6; ++a) {
            double* row = c->S + (size_t)(6 * cams[i] + a) * n + 6 * cams[j];
            for (int b = 0; b < 6; ++b) row[b] -= B[a * 6 + b];
          }
          __atomic_clear(lk, __ATOMIC_RELEASE);
        }
      }
    }
#pragma omp critical
    for (int i = 0; i < n; ++i) rhs_add[i] += rloc[i];
    free(rloc); free(cams); free(W);
  }
  /* stash the W Vinv g_p part in the yc buffer slot of the reduced vector: we keep it in
   * c->rhs for now (the caller adds -gc after the all-reduce) */
  memcpy(c->rhs, rhs_add, sizeof(double) * (size_t)n);
  free(rhs_add);
  return !bad;
}

/* Build, all-reduce and solve the reduced camera system for trust-region radius `radius`,
 * then back-substitute.  On success c->yc / c->yp hold the step in scaled coordinates and
 * *model_cost_change is set.  Returns 1 ok, 0 linear-solver failure, <0 comm error. */
static int compute_step(ctx_t* c, double radius, const mpsfm_ba_options* o, double* model_cost_change,
                        double* S_out, double* rhs_out) {
  const int n = c->n;
  /* the diagonal blocks of S currently hold U (from linearize); keep a copy so a rejected step
   * can rebuild with another radius without re-linearising */
  double* Ukeep = (double*)malloc(sizeof(double) * 36 * (size_t)(c->ncv + 1));
  for (int s = 0; s < c->ncv; ++s)
    for (int a = 0; a < 6; ++a)
      memcpy(Ukeep + 36 * s + 6 * a, c->S + (size_t)(6 * s + a) * n + 6 * s, sizeof(double) * 6);
  double* Vkeep = (double*)malloc(sizeof(double) * 6 * (size_t)(c->np + 1));
  memcpy(Vkeep, c->V, sizeof(double) * 6 * (size_t)c->np);
  double* gckeep = (double*)malloc(sizeof(double) * (size_t)(2 * n + 8));
  memcpy(gckeep, c->gc, sizeof(double) * (size_t)(2 * n + 8));

  int ok = schur_reduce(c, radius, o->min_lm_diagonal, o->max_lm_diagonal);
  /* reduced vector rides in the gc slot for the all-reduce: gc_slot := rhs_add - gc_local */
  if (c->allreduce) {
    /* in sharded mode U, gc and diagU were already globally summed by the caller after
     * linearize(); only this shard's Schur part may be reduced now.  To keep one code path the
     * caller passes local-only U/gc here and we reduce everything at once. */
  }
  for (int i = 0; i < n; ++i) c->gc[i] = c->rhs[i] - c->gc[i];
  c->sc[1] = ok ? 0.0 : 1.0;
  int comm = do_allreduce(c, c->red, c->red_count);
  if (comm) { free(Ukeep); free(Vkeep); free(gckeep); return MPSFM_ECOMM; }
  ok = (c->sc[1] == 0.0);
  memcpy(c->rhs, c->gc, sizeof(double) * (size_t)n);
  /* LM damping of the camera block: D_c = clamp(diag U) / radius (diagU is now global) */
  for (int i = 0; i < n; ++i) {
    const double d = fmin(fmax(c->diagU[i], o->min_lm_diagonal), o->max_lm_diagonal) / radius;
    c->S[(size_t)i * n + i] += d;
  }
  /* symmetrise (upper block triangle was accumulated) */
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const int bi = i / 6, bj = j / 6;
      if (bi == bj) continue; /* diagonal blocks are full */
      c->S[(size_t)j * n + i] = c->S[(size_t)i * n + j];
    }
  if (S_out) memcpy(S_out, c->S, sizeof(double) * (size_t)n * n);
  if (rhs_out) memcpy(rhs_out, c->rhs, sizeof(double) * (size_t)n);
  if (ok && n > 0) {
    memcpy(c->yc, c->rhs, sizeof(double) * (size_t)n);
    if (!dense_cholesky(c->S, n)) ok = 0;
    else dense_chol_solve(c->S, n, c->yc);
    for (int i = 0; i < n; ++i) if (!isfinite(c->yc[i])) ok = 0;
  }
  double mcc = 0.0;
  if (ok) {
    /* back-substitution and model cost change = -sum m.(r + m/2), m = Jc yc + Jp yp */
#pragma omp parallel for schedule(dynamic, 512) reduction(+ : mcc)
    for (int p = 0; p < c->np; ++p) {
      double acc[3] = {0, 0, 0};
      double* yp = c->yp + 3 * p;
      const int64_t b0 = c->pt_start[p], b1 = c->pt_start[p + 1];
      if (c->pvar[p]) {
        for (int64_t k = b0; k < b1; ++k) {
          const int slot = c->cam_sys[c->blk_cam[k]];
          if (slot < 0) continue;
          const double* jc = c->Jc + 12 * k;
          const double* jp = c->Jp + 6 * k;
          const double* y = c->yc + 6 * slot;
          for (int a = 0; a < 2; ++a) {
            double u = 0.0;
            for (int i = 0; i < 6; ++i) u += jc[a * 6 + i] * y[i];
            acc[0] += jp[a * 3] * u; acc[1] += jp[a * 3 + 1] * u; acc[2] += jp[a * 3 + 2] * u;
          }
        }
        const double* g = c->gp + 3 * p;
        double tmp[3] = {-(g[0] + acc[0]), -(g[1] + acc[1]), -(g[2] + acc[2])};
        sym3_mul(c->V + 6 * p, tmp, yp);
      } else {
        yp[0] = yp[1] = yp[2] = 0.0;
      }
      for (int64_t k = b0; k < b1; ++k) {
        const int slot = c->cam_sys[c->blk_cam[k]];
        const double* jc = c->Jc + 12 * k;
        const double* jp = c->Jp + 6 * k;
        const double* r = c->rr + 2 * k;
        for (int a = 0; a < 2; ++a) {
          double m = jp[a * 3] * yp[0] + jp[a * 3 + 1] * yp[1] + jp[a * 3 + 2] * yp[2];
          if (slot >= 0) {
            const double* y = c->yc + 6 * slot;
            for (int i = 0; i < 6; ++i) m += jc[a * 6 + i] * y[i];
          }
          mcc -= m * (r[a] + 0.5 * m);
        }
      }
    }
  }
  double red2[2] = {mcc, 0.0};
  if (do_allreduce(c, red2, 2)) { free(Ukeep); free(Vkeep); free(gckeep); return MPSFM_ECOMM; }
  *model_cost_change = red2[0];
  /* restore the linearisation products for a possible retry with another radius */
  memset(c->S, 0, sizeof(double) * (size_t)n * n);
  for (int s = 0; s < c->ncv; ++s)
    for (int a = 0; a < 6; ++a)
      memcpy(c->S + (size_t)(6 * s + a) * n + 6 * s, Ukeep + 36 * s + 6 * a, sizeof(double) * 6);
  /* (V+D)^-1 of this radius is still needed by nobody after back-substitution */
  memcpy(c->V, Vkeep, sizeof(double) * 6 * (size_t)c->np);
  memcpy(c->gc, gckeep, sizeof(double) * (size_t)(2 * n + 8));
  free(Ukeep); free(Vkeep); free(gckeep);
  return ok;
}

/* x [+] delta for cameras and points; delta = scale .* y */
static void apply_step(ctx_t* c, const double* q, const double* t, const double* pts, double* q2, double* t2,
                       double* pts2) {
  for (int i = 0; i < c->nc; ++i) {
    const int slot = c->cam_sys[i];
    if (slot < 0) {
      memcpy(q2 + 4 * i, q + 4 * i, sizeof(double) * 4);
      memcpy(t2 + 3 * i, t + 3 * i, sizeof(double) * 3);
      continue;
    }
    double d[6];
    for (int k = 0; k < 6; ++k) d[k] = c->cs[6 * i + k] * c->yc[6 * slot + k];
    quat_plus(q + 4 * i, d, q2 + 4 * i);
    for (int k = 0; k < 3; ++k) t2[3 * i + k] = t[3 * i + k] + d[3 + k];
  }
#pragma omp parallel for schedule(static)
  for (int p = 0; p < c->np; ++p)
    for (int k = 0; k < 3; ++k)
      pts2[3 * p + k] = pts[3 * p + k] + (c->pvar[p] ? c->ps[3 * p + k] * c->yp[3 * p + k] : 0.0);
}

/* squared ambient norms; cameras are replicated on every shard, points are local */
static double cam_sqnorm(ctx_t* c, const double* q, const double* t) {
  double s = 0.0;
  for (int i = 0; i < c->nc; ++i) {
    if (c->cam_sys[i] < 0) continue;
    for (int k = 0; k < 4; ++k) s += q[4 * i + k] * q[4 * i + k];
    for (int k = 0; k < 3; ++k) s += t[3 * i + k] * t[3 * i + k];
  }
  return s;
}
static double cam_sqdiff(ctx_t* c, const double* q, const double* t, const double* q2, const double* t2) {
  double s = 0.0;
  for (int i = 0; i < c->nc; ++i) {
    if (c->cam_sys[i] < 0) continue;
    for (int k = 0; k < 4; ++k) { const double d = q[4 * i + k] - q2[4 * i + k]; s += d * d; }
    for (int k = 0; k < 3; ++k) { const double d = t[3 * i + k] - t2[3 * i + k]; s += d * d; }
  }
  return s;
}
static double pts_sqnorm(ctx_t* c, const double* pts) {
  double s = 0.0;
  for (int p = 0; p < c->np; ++p)
    if (c->pvar[p]) for (int k = 0; k < 3; ++k) s += pts[3 * p + k] * pts[3 * p + k];
  return s;
}
static double pts_sqdiff(ctx_t* c, const double* a, const double* b) {
  double s = 0.0;
  for (int p = 0; p < c->np; ++p)
    if (c->pvar[p]) for (int k = 0; k < 3; ++k) { const double d = a[3 * p + k] - b[3 * p + k]; s += d * d; }
  return s;
}

/* gradient max norm |x - Plus(x, -g)|_inf in the ambient space; g = unscaled gradient.
 * gc here must be the GLOBAL scaled camera gradient; point part is local (max-reduced by
 * summing is wrong, so the caller all-reduces via a max trick: we return both parts). */
static double grad_max_norm_cams(ctx_t* c, const double* q, const double* gc_scaled) {
  double m = 0.0;
  for (int i = 0; i < c->nc; ++i) {
    const int slot = c->cam_sys[i];
    if (slot < 0) continue;
    double g[6];
    for (int k = 0; k < 6; ++k) {
      const double s = c->cs[6 * i + k];
      g[k] = s > 0.0 ? -gc_scaled[6 * slot + k] / s : 0.0;
    }
    double q2[4];
    quat_plus(q + 4 * i, g, q2);
    for (int k = 0; k < 4; ++k) m = fmax(m, fabs(q2[k] - q[4 * i + k]));
    for (int k = 0; k < 3; ++k) m = fmax(m, fabs(g[3 + k]));
  }
  return m;
}
static double grad_max_norm_pts(ctx_t* c) {
  double m = 0.0;
  for (int p = 0; p < c->np; ++p) {
    if (!c->pvar[p]) continue;
    for (int k = 0; k < 3; ++k) {
      const double s = c->ps[3 * p + k];
      if (s > 0.0) m = fmax(m, fabs(c->gp[3 * p + k] / s));
    }
  }
  return m;
}

ORACLE_API void oracle_default_options(mpsfm_ba_options* o) {
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
}

/* Jacobi scaling from the Jacobian at the current point: 1/(1+sqrt(col norm^2)). */
static int compute_jacobi_scaling(ctx_t* c, const double* q, const double* t, const double* pts, int enable) {
  for (int i = 0; i < c->nc * 6; ++i) c->cs[i] = c->cmask[i];
  for (int p = 0; p < c->np; ++p) for (int k = 0; k < 3; ++k) c->ps[3 * p + k] = c->pvar[p] ? 1.0 : 0.0;
  if (!enable) return 1;
  if (!linearize(c, q, t, pts)) return 0;
  /* camera column norms need the global sum */
  c->sc[1] = 0.0;
  if (do_allreduce(c, c->gc, 2 * (int64_t)c->n + 8)) return -1;
  for (int i = 0; i < c->nc; ++i) {
    const int slot = c->cam_sys[i];
    if (slot < 0) continue;
    for (int k = 0; k < 6; ++k) c->cs[6 * i + k] = c->cmask[6 * i + k] / (1.0 + sqrt(c->diagU[6 * slot + k]));
  }
  for (int p = 0; p < c->np; ++p) {
    if (!c->pvar[p]) continue;
    c->ps[3 * p] = 1.0 / (1.0 + sqrt(c->V[6 * p]));
    c->ps[3 * p + 1] = 1.0 / (1.0 + sqrt(c->V[6 * p + 3]));
    c->ps[3 * p + 2] = 1.0 / (1.0 + sqrt(c->V[6 * p + 5]));
  }
  return 1;
}

static int oracle_solve_impl(const mpsfm_ba_problem* P, mpsfm_ba_state* st, const mpsfm_ba_options* o,
                             mpsfm_ba_summary* sum) {
  if (!check_problem(P) || !st || !o || !sum) return MPSFM_EINVAL;
  const double t_begin = now_s();
  memset(sum, 0, sizeof(*sum));
  ctx_t C;
  ctx_t* c = &C;
  int rc = ctx_build(c, P, o->allreduce, o->allreduce_user);
  if (rc) { ctx_free(c); return rc; }
  const int nc = c->nc, np = c->np;
  double* q = st->cam_quat_xyzw; double* t = st->cam_t; double* pts = st->pts;
  double* q2 = (double*)malloc(sizeof(double) * 4 * (size_t)(nc + 1));
  double* t2 = (double*)malloc(sizeof(double) * 3 * (size_t)(nc + 1));
  double* pts2 = (double*)malloc(sizeof(double) * 3 * (size_t)(np + 1));
  int64_t cnt[2] = {0, 0};
  double tot[4] = {(double)(P->n_obs + P->n_dobs), (double)c->nblk, 0, 0};
  for (int p = 0; p < np; ++p) tot[2] += c->pvar[p];
  do_allreduce(c, tot, 4);
  sum->num_residual_blocks = (int64_t)tot[0];
  const int64_t nblk_global = (int64_t)tot[1];
  const double nvarpts_global = tot[2];
  sum->reduced_dim = c->n;
  double fixed = eval_fixed_cost(c, q, t, pts, NULL);
  { double f[1] = {fixed}; do_allreduce(c, f, 1); fixed = f[0]; }
  sum->fixed_cost = fixed;
  int term = -1;
  int ret = 0;

  if (c->n == 0 && nvarpts_global == 0.0) {
    double cost = eval_cost(c, q, t, pts, NULL);
    { double f[1] = {cost}; do_allreduce(c, f, 1); cost = f[0]; }
    sum->initial_cost = sum->final_cost = cost + fixed;
    sum->termination = MPSFM_TERM_NO_VARIABLES;
    goto done;
  }

  /* iteration 0 */
  {
    int js = compute_jacobi_scaling(c, q, t, pts, o->jacobi_scaling);
    if (js < 0) { ret = MPSFM_ECOMM; goto done; }
    if (js == 0) { ret = MPSFM_ENUMERIC; goto done; }
  }
  if (!linearize(c, q, t, pts)) { ret = MPSFM_ENUMERIC; goto done; }
  cnt[1]++;
  double x_cost;
  {
    /* cost and camera gradient need the global sums; S diag/gc/diagU are reduced inside
     * compute_step together with the Schur part, so reduce a copy of the scalars + gc here */
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(c->n + 2));
    memcpy(tmp, c->gc, sizeof(double) * (size_t)c->n);
    tmp[c->n] = c->sc[0];
    tmp[c->n + 1] = isfinite(c->sc[0]) ? 0.0 : 1.0;
    if (do_allreduce(c, tmp, c->n + 2)) { free(tmp); ret = MPSFM_ECOMM; goto done; }
    x_cost = tmp[c->n];
    double gmax = grad_max_norm_cams(c, q, tmp);
    free(tmp);
    double gm2[1] = {grad_max_norm_pts(c)};
    /* max over shards via the sum hook: not available; the point part only matters for the
     * 1e-10 gradient tolerance, use the local value on every shard combined through a sum of
     * indicator (>tol) below */
    gmax = fmax(gmax, gm2[0]);
    double ind[1] = {gmax > o->gradient_tolerance ? 1.0 : 0.0};
    do_allreduce(c, ind, 1);
    sum->initial_cost = x_cost + fixed;
    sum->trace_cost[0] = x_cost + fixed; sum->trace_radius[0] = o->initial_trust_region_radius;
    sum->trace_accepted[0] = 1; sum->trace_len = 1;
    if (ind[0] == 0.0) { term = MPSFM_TERM_GRADIENT_TOLERANCE; }
  }
  double x_norm;
  {
    double v[1] = {pts_sqnorm(c, pts)};
    do_allreduce(c, v, 1);
    x_norm = sqrt(v[0] + cam_sqnorm(c, q, t));
  }
  double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
  int iter = 0, invalid_run = 0;
  while (term < 0) {
    if (iter >= o->max_num_iterations) { term = MPSFM_TERM_MAX_ITERATIONS; break; }
    if (radius <= o->min_trust_region_radius) { term = MPSFM_TERM_MIN_RADIUS; break; }
    ++iter;
    double mcc = 0.0;
    int ok = compute_step(c, radius, o, &mcc, NULL, NULL);
    if (ok < 0) { ret = MPSFM_ECOMM; break; }
    const int step_valid = ok && (mcc > 0.0);
    if (!step_valid) {
      ++invalid_run;
      sum->num_unsuccessful_steps++;
      if (invalid_run >= o->max_num_consecutive_invalid_steps) { term = MPSFM_TERM_INVALID_STEPS; }
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (sum->trace_len < MPSFM_MAX_TRACE) {
        sum->trace_cost[sum->trace_len] = x_cost + fixed; sum->trace_radius[sum->trace_len] = radius;
        sum->trace_accepted[sum->trace_len] = 0; sum->trace_len++;
      }
      continue;
    }
    invalid_run = 0;
    apply_step(c, q, t, pts, q2, t2, pts2);
    double cand = eval_cost(c, q2, t2, pts2, NULL);
    cnt[0]++;
    double red[3] = {isfinite(cand) ? cand : 0.0, isfinite(cand) ? 0.0 : 1.0, pts_sqdiff(c, pts, pts2)};
    if (do_allreduce(c, red, 3)) { ret = MPSFM_ECOMM; break; }
    cand = red[1] > 0.0 ? DBL_MAX : red[0];
    const double step_norm = sqrt(red[2] + cam_sqdiff(c, q, t, q2, t2));
    if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
      term = MPSFM_TERM_PARAMETER_TOLERANCE; break;
    }
    const double cost_change = x_cost - cand;
    if (fabs(cost_change) <= o->function_tolerance * x_cost) { term = MPSFM_TERM_FUNCTION_TOLERANCE; break; }
    const double rel = cost_change / mcc;
    if (rel > o->min_relative_decrease) {
      memcpy(q, q2, sizeof(double) * 4 * (size_t)nc);
      memcpy(t, t2, sizeof(double) * 3 * (size_t)nc);
      memcpy(pts, pts2, sizeof(double) * 3 * (size_t)np);
      { double v[1] = {pts_sqnorm(c, pts)}; do_allreduce(c, v, 1); x_norm = sqrt(v[0] + cam_sqnorm(c, q, t)); }
      if (!linearize(c, q, t, pts)) { ret = MPSFM_ENUMERIC; break; }
      cnt[1]++;
      double* tmp = (double*)malloc(sizeof(double) * (size_t)(c->n + 2));
      memcpy(tmp, c->gc, sizeof(double) * (size_t)c->n);
      tmp[c->n] = c->sc[0]; tmp[c->n + 1] = 0.0;
      if (do_allreduce(c, tmp, c->n + 2)) { free(tmp); ret = MPSFM_ECOMM; break; }
      x_cost = tmp[c->n];
      double gmax = fmax(grad_max_norm_cams(c, q, tmp), grad_max_norm_pts(c));
      free(tmp);
      double ind[1] = {gmax > o->gradient_tolerance ? 1.0 : 0.0};
      do_allreduce(c, ind, 1);
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3.0));
      radius = fmin(o->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      sum->num_successful_steps++;
      if (sum->trace_len < MPSFM_MAX_TRACE) {
        sum->trace_cost[sum->trace_len] = x_cost + fixed; sum->trace_radius[sum->trace_len] = radius;
        sum->trace_accepted[sum->trace_len] = 1; sum->trace_len++;
      }
      if (ind[0] == 0.0) { term = MPSFM_TERM_GRADIENT_TOLERANCE; }
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
      if (sum->trace_len < MPSFM_MAX_TRACE) {
        sum->trace_cost[sum->trace_len] = x_cost + fixed; sum->trace_radius[sum->trace_len] = radius;
        sum->trace_accepted[sum->trace_len] = 0; sum->trace_len++;
      }
    }
  }
  if (ret == 0) {
    sum->final_cost = x_cost + fixed;
    sum->num_iterations = iter;
    sum->termination = term;
    sum->final_radius = radius;
  }
done:
  sum->num_jacobian_evals = cnt[1];
  sum->num_residual_evals = nblk_global * (cnt[0] + cnt[1]);
  sum->time_total_s = now_s() - t_begin;
  free(q2); free(t2); free(pts2);
  ctx_free(c);
  return ret;
}

ORACLE_API int oracle_ba_solve(const mpsfm_ba_problem* P, mpsfm_ba_state* st, const mpsfm_ba_options* o,
                               mpsfm_ba_summary* sum) {
  return oracle_solve_impl(P, st, o, sum);
}

/* cost split at a state: out[0] reprojection, out[1] depth (both include fixed blocks) */
ORACLE_API int oracle_ba_eval_cost(const mpsfm_ba_problem* P, const mpsfm_ba_state* st, double* out2) {
  if (!check_problem(P)) return MPSFM_EINVAL;
  ctx_t C;
  int rc = ctx_build(&C, P, NULL, NULL);
  if (rc) { ctx_free(&C); return rc; }
  double a[2], b[2];
  eval_cost(&C, st->cam_quat_xyzw, st->cam_t, st->pts, a);
  eval_fixed_cost(&C, st->cam_quat_xyzw, st->cam_t, st->pts, b);
  out2[0] = a[0] + b[0]; out2[1] = a[1] + b[1];
  ctx_free(&C);
  return 0;
}

/* Reduced camera system at a state for a given radius (Jacobi scaling computed at that same
 * state, as in iteration 0).  S: n*n row-major, rhs: n, cam_scale: [n_cams][6],
 * pt_scale: [n_pts][3] (either scale pointer may be NULL).  Returns n or <0. */
ORACLE_API int oracle_reduced_system(const mpsfm_ba_problem* P, const mpsfm_ba_state* st, double radius,
                                     int jacobi, double* S, double* rhs, double* cam_scale, double* pt_scale,
                                     double* yc_out, double* yp_out, double* mcc_out) {
  if (!check_problem(P)) return MPSFM_EINVAL;
  ctx_t C;
  int rc = ctx_build(&C, P, NULL, NULL);
  if (rc) { ctx_free(&C); return rc; }
  mpsfm_ba_options o;
  oracle_default_options(&o);
  if (compute_jacobi_scaling(&C, st->cam_quat_xyzw, st->cam_t, st->pts, jacobi) <= 0) { ctx_free(&C); return MPSFM_ENUMERIC; }
  if (!linearize(&C, st->cam_quat_xyzw, st->cam_t, st->pts)) { ctx_free(&C); return MPSFM_ENUMERIC; }
  double mcc = 0.0;
  int ok = compute_step(&C, radius, &o, &mcc, S, rhs);
  if (cam_scale) memcpy(cam_scale, C.cs, sizeof(double) * 6 * (size_t)C.nc);
  if (pt_scale) memcpy(pt_scale, C.ps, sizeof(double) * 3 * (size_t)C.np);
  if (yc_out) memcpy(yc_out, C.yc, sizeof(double) * (size_t)C.n);
  if (yp_out) memcpy(yp_out, C.yp, sizeof(double) * 3 * (size_t)C.np);
  if (mcc_out) *mcc_out = mcc;
  const int n = C.n;
  ctx_free(&C);
  return ok > 0 ? n : MPSFM_ENUMERIC;
}

ORACLE_API int oracle_reduced_dim(const mpsfm_ba_problem* P) {
  if (!check_problem(P)) return MPSFM_EINVAL;
  ctx_t C;
  int rc = ctx_build(&C, P, NULL, NULL);
  const int n = C.n;
  ctx_free(&C);
  return rc ? rc : n;
}

/* Point covariances (reference bundle_adjustment.py:244-261 -> pycolmap.estimate_ba_covariance
 * with params=POINTS): reprojection-only problem, trivial loss scaled by the magnitude,
 * cov_j = (sum_i k Jp_i^T Jp_i)^-1, conditioned on every other variable.  Points with no
 * observation (or a singular block) get NaN. */
ORACLE_API int oracle_point_covs(const mpsfm_ba_problem* P, const mpsfm_ba_state* st, double* covs) {
  if (!check_problem(P)) return MPSFM_EINVAL;
  const int np = P->n_pts;
  double* H = (double*)calloc((size_t)np * 6 + 6, sizeof(double));
  for (int64_t i = 0; i < P->n_obs; ++i) {
    const int cam = P->obs_cam[i], pt = P->obs_pt[i];
    double R[9], r[2], Jc[12], Jp[6];
    quat_to_R(st->cam_quat_xyzw + 4 * cam, R);
    block_eval(0, R, st->cam_t + 3 * cam, P->cam_intr + 4 * P->cam_intr_idx[cam], st->pts + 3 * pt,
               P->obs_xy + 2 * i, 0.0, r, Jc, Jp);
    const double k = P->reproj_loss_magnitude;
    double* h = H + 6 * pt;
    for (int a = 0; a < 2; ++a) {
      const double* jp = Jp + 3 * a;
      h[0] += k * jp[0] * jp[0]; h[1] += k * jp[0] * jp[1]; h[2] += k * jp[0] * jp[2];
      h[3] += k * jp[1] * jp[1]; h[4] += k * jp[1] * jp[2]; h[5] += k * jp[2] * jp[2];
    }
  }
  for (int p = 0; p < np; ++p) {
    double Vi[6];
    double* o = covs + 9 * p;
    if (!spd3_inverse(H + 6 * p, Vi)) { for (int k = 0; k < 9; ++k) o[k] = NAN; continue; }
    o[0] = Vi[0]; o[1] = Vi[1]; o[2] = Vi[2];
    o[3] = Vi[1]; o[4] = Vi[3]; o[5] = Vi[4];
    o[6] = Vi[2]; o[7] = Vi[4]; o[8] = Vi[5];
  }
  free(H);
  return 0;
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
