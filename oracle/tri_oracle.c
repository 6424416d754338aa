/*
 * tri_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of the per-track triangulation numerics that the reference reaches through
 * pycolmap.IncrementalTriangulator / ObservationManager (call sites
 * mpsfm/sfm/mapper/triangulator.py:48,53-55,123-128; mapper/base.py:686-797;
 * mpsfm/utils/geometry.py:54-75).  COLMAP (fork, unpinned) is not in the reference tree:
 * PARITY UNPINNED; the formulas follow COLMAP's published TriangulateMultiViewPoint,
 * CalculateTriangulationAngle and HasPointPositiveDepth.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../include/mpsfm_hip.h"
#define ORACLE_API __attribute__((visibility("default")))

static void q2R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* cyclic Jacobi eigen-decomposition of a symmetric 4x4; returns eigenvector of the smallest
 * eigenvalue in v */
static void sym4_min_eigvec(double A[16], double v[4]) {
  double Q[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i * 4 + j] * A[i * 4 + j];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p * 4 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 4 + q] - A[p * 4 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = A[k * 4 + p], akq = A[k * 4 + q];
          A[k * 4 + p] = c * akp - s * akq; A[k * 4 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[p * 4 + k], aqk = A[q * 4 + k];
          A[p * 4 + k] = c * apk - s * aqk; A[q * 4 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double qkp = Q[k * 4 + p], qkq = Q[k * 4 + q];
          Q[k * 4 + p] = c * qkp - s * qkq; Q[k * 4 + q] = s * qkp + c * qkq;
        }
      }
  }
  int m = 0;
  for (int i = 1; i < 4; ++i) if (A[i * 4 + i] < A[m * 4 + m]) m = i;
  for (int k = 0; k < 4; ++k) v[k] = Q[k * 4 + m];
}

ORACLE_API int oracle_triangulate_tracks(const mpsfm_tracks* T, double* xyz) {
  for (int t = 0; t < T->n_tracks; ++t) {
    double A[16];
    memset(A, 0, sizeof(A));
    for (int64_t e = T->track_start[t]; e < T->track_start[t + 1]; ++e) {
      const int cam = T->el_cam[e];
      const double* K = T->cam_intr + 4 * T->cam_intr_idx[cam];
      double R[9], P[12];
      q2R(T->cam_quat_xyzw + 4 * cam, R);
      for (int i = 0; i < 3; ++i) { P[i * 4] = R[i * 3]; P[i * 4 + 1] = R[i * 3 + 1]; P[i * 4 + 2] = R[i * 3 + 2]; P[i * 4 + 3] = T->cam_t[3 * cam + i]; }
      double x[3] = {(T->el_xy[2 * e] - K[2]) / K[0], (T->el_xy[2 * e + 1] - K[3]) / K[1], 1.0};
      const double nrm = sqrt(x[0] * x[0] + x[1] * x[1] + 1.0);
      x[0] /= nrm; x[1] /= nrm; x[2] /= nrm;
      /* term = P - x x^T P */
      double xtP[4], term[12];
      for (int j = 0; j < 4; ++j) xtP[j] = x[0] * P[j] + x[1] * P[4 + j] + x[2] * P[8 + j];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) term[i * 4 + j] = P[i * 4 + j] - x[i] * xtP[j];
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
        A[i * 4 + j] += term[i] * term[j] + term[4 + i] * term[4 + j] + term[8 + i] * term[8 + j];
    }
    double v[4];
    sym4_min_eigvec(A, v);
    xyz[3 * t] = v[0] / v[3]; xyz[3 * t + 1] = v[1] / v[3]; xyz[3 * t + 2] = v[2] / v[3];
  }
  return 0;
}

ORACLE_API int oracle_filter_tracks(const mpsfm_tracks* T, const double* xyz, double* max_angle, double* sq_err,
                                    uint8_t* front) {
  for (int t = 0; t < T->n_tracks; ++t) {
    const double* X = xyz + 3 * t;
    const int64_t e0 = T->track_start[t], e1 = T->track_start[t + 1];
    double best = 0.0;
    for (int64_t e = e0; e < e1; ++e) {
      const int cam = T->el_cam[e];
      const double* K = T->cam_intr + 4 * T->cam_intr_idx[cam];
      double R[9];
      q2R(T->cam_quat_xyzw + 4 * cam, R);
      const double* tt = T->cam_t + 3 * cam;
      const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + tt[0];
      const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + tt[1];
      const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + tt[2];
      if (front) front[e] = zc >= 2.220446049250313e-16;
      if (sq_err) {
        const double du = K[0] * xc / zc + K[2] - T->el_xy[2 * e];
        const double dv = K[1] * yc / zc + K[3] - T->el_xy[2 * e + 1];
        sq_err[e] = du * du + dv * dv;
      }
      if (max_angle) {
        /* projection centre C = -R^T t */
        double C1[3] = {-(R[0] * tt[0] + R[3] * tt[1] + R[6] * tt[2]), -(R[1] * tt[0] + R[4] * tt[1] + R[7] * tt[2]),
                        -(R[2] * tt[0] + R[5] * tt[1] + R[8] * tt[2])};
        for (int64_t f = e + 1; f < e1; ++f) {
          const int cam2 = T->el_cam[f];
          double R2[9];
          q2R(T->cam_quat_xyzw + 4 * cam2, R2);
          const double* t2 = T->cam_t + 3 * cam2;
          double C2[3] = {-(R2[0] * t2[0] + R2[3] * t2[1] + R2[6] * t2[2]), -(R2[1] * t2[0] + R2[4] * t2[1] + R2[7] * t2[2]),
                          -(R2[2] * t2[0] + R2[5] * t2[1] + R2[8] * t2[2])};
          double b2 = 0, r1 = 0, r2 = 0;
          for (int k = 0; k < 3; ++k) {
            b2 += (C1[k] - C2[k]) * (C1[k] - C2[k]);
            r1 += (X[k] - C1[k]) * (X[k] - C1[k]);
            r2 += (X[k] - C2[k]) * (X[k] - C2[k]);
          }
          const double den = 2.0 * sqrt(r1 * r2);
          double ang = 0.0;
          if (den != 0.0) {
            double cs = (r1 + r2 - b2) / den;
            cs = cs > 1.0 ? 1.0 : (cs < -1.0 ? -1.0 : cs);
            ang = fabs(acos(cs));
            ang = ang < M_PI - ang ? ang : M_PI - ang;
          }
          if (ang > best) best = ang;
        }
      }
    }
    if (max_angle) max_angle[t] = best;
  }
  return 0;
}
