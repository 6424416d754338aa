// Per-track triangulation arithmetic shared by the batch kernels (tri_kernels.hip) and the track-graph engine
// (triangulator.hip): COLMAP 3.11 semantics of TriangulatePoint / TriangulateMultiViewPoint /
// CalculateTriangulationAngle / CalculateNormalizedAngularError / CalculateSquaredReprojectionError /
// HasPointPositiveDepth and of LORANSAC<TriangulationEstimator, TriangulationEstimator, InlierSupportMeasurer,
// CombinationSampler> as EstimateTriangulation runs it (reached in the reference through
// pycolmap.IncrementalTriangulator, mpsfm/sfm/mapper/triangulator.py:32-48, 88-123).  The fork's source is not in the
// reference tree: parity unpinned, semantics restated from upstream COLMAP.
#pragma once
#include "common.h"

namespace mpsfm {

// smallest eigenvector of a symmetric 4x4 by cyclic Jacobi rotations
__host__ __device__ inline void sym4_min_eigvec(double A[4][4], double v[4]) {
  double Q[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double qkp = Q[k][p], qkq = Q[k][q];
          Q[k][p] = c * qkp - s * qkq; Q[k][q] = s * qkp + c * qkq;
        }
      }
  }
  double best = A[0][0];
  v[0] = Q[0][0]; v[1] = Q[1][0]; v[2] = Q[2][0]; v[3] = Q[3][0];
  for (int i = 1; i < 4; ++i)
    if (A[i][i] < best) { best = A[i][i]; v[0] = Q[0][i]; v[1] = Q[1][i]; v[2] = Q[2][i]; v[3] = Q[3][i]; }
}

// One observation as the estimator sees it: 3x4 cam_from_world (row-major), projection centre, normalised image point
struct TriView {
  double P[12];
  double C[3];
  double xn[2];
  double xy[2];  // the pixel measurement and the PINHOLE intrinsics: the reprojection residual of CompleteImage needs them
  double K[4];
};

__host__ __device__ inline void tri_make_view(const double* R, const double* t, const double* K, const double* xy, TriView& v) {
  for (int i = 0; i < 3; ++i) { v.P[4 * i] = R[3 * i]; v.P[4 * i + 1] = R[3 * i + 1]; v.P[4 * i + 2] = R[3 * i + 2]; v.P[4 * i + 3] = t[i]; }
  for (int k = 0; k < 3; ++k) v.C[k] = -(R[k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]);
  v.xn[0] = (xy[0] - K[2]) / K[0];
  v.xn[1] = (xy[1] - K[3]) / K[1];
  v.xy[0] = xy[0]; v.xy[1] = xy[1];
  for (int k = 0; k < 4; ++k) v.K[k] = K[k];
}

__host__ __device__ inline double tri_depth(const double* P, const double* X) { return P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11]; }
__host__ __device__ inline bool tri_positive_depth(const double* P, const double* X) { return tri_depth(P, X) >= 2.220446049250313e-16; }

__host__ __device__ inline double tri_angle(const double* C1, const double* C2, const double* X) {
  double b2 = 0, r1 = 0, r2 = 0;
  for (int k = 0; k < 3; ++k) {
    b2 += (C1[k] - C2[k]) * (C1[k] - C2[k]);
    r1 += (X[k] - C1[k]) * (X[k] - C1[k]);
    r2 += (X[k] - C2[k]) * (X[k] - C2[k]);
  }
  const double den = 2.0 * sqrt(r1 * r2);
  if (den == 0.0) return 0.0;
  double cs = (r1 + r2 - b2) / den;
  cs = cs > 1.0 ? 1.0 : (cs < -1.0 ? -1.0 : cs);
  const double ang = fabs(acos(cs));
  return fmin(ang, M_PI - ang);
}

// angle between the viewing ray of the normalised point and the ray to X in the camera frame
__host__ __device__ inline double tri_angular_error(const double* xn, const double* X, const double* P) {
  const double a[3] = {xn[0], xn[1], 1.0};
  const double b[3] = {P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3], P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7], tri_depth(P, X)};
  const double na = sqrt(a[0] * a[0] + a[1] * a[1] + 1.0), nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  double cs = (a[0] * b[0] + a[1] * b[1] + b[2]) / (na * nb);
  cs = cs > 1.0 ? 1.0 : (cs < -1.0 ? -1.0 : cs);
  return acos(cs);
}

// squared reprojection error in pixels; DBL_MAX behind the camera (COLMAP CalculateSquaredReprojectionError)
__host__ __device__ inline double tri_sq_reproj_error(const double* xy, const double* X, const double* P, const double* K) {
  const double z = tri_depth(P, X);
  if (z < 2.220446049250313e-16) return DBL_MAX;
  const double x = (P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3]) / z, y = (P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7]) / z;
  const double du = K[0] * x + K[2] - xy[0], dv = K[1] * y + K[3] - xy[1];
  return du * du + dv * dv;
}

__host__ __device__ inline void tri_two_view(const TriView& a, const TriView& b, double* X) {
  double Arow[4][4];
  for (int k = 0; k < 4; ++k) {
    Arow[0][k] = a.xn[0] * a.P[8 + k] - a.P[k];
    Arow[1][k] = a.xn[1] * a.P[8 + k] - a.P[4 + k];
    Arow[2][k] = b.xn[0] * b.P[8 + k] - b.P[k];
    Arow[3][k] = b.xn[1] * b.P[8 + k] - b.P[4 + k];
  }
  double M[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) M[i][j] = Arow[0][i] * Arow[0][j] + Arow[1][i] * Arow[1][j] + Arow[2][i] * Arow[2][j] + Arow[3][i] * Arow[3][j];
  double v[4];
  sym4_min_eigvec(M, v);
  X[0] = v[0] / v[3]; X[1] = v[1] / v[3]; X[2] = v[2] / v[3];
}

// accumulates one view into the 4x4 normal matrix of TriangulateMultiViewPoint
__host__ __device__ inline void tri_multi_accumulate(const TriView& w, double A[4][4]) {
  double x[3] = {w.xn[0], w.xn[1], 1.0};
  const double nrm = sqrt(x[0] * x[0] + x[1] * x[1] + 1.0);
  x[0] /= nrm; x[1] /= nrm; x[2] /= nrm;
  double xtP[4], term[3][4];
  for (int j = 0; j < 4; ++j) xtP[j] = x[0] * w.P[j] + x[1] * w.P[4 + j] + x[2] * w.P[8 + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) term[i][j] = w.P[4 * i + j] - x[i] * xtP[j];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) A[i][j] += term[0][i] * term[0][j] + term[1][i] * term[1][j] + term[2][i] * term[2][j];
}

constexpr int kTriMaxViews = 64;  // longest candidate track the GPU batch takes (fixed scratch per thread); longer ones are
                                  // estimated on the host with the same function and heap scratch (tri_ransac_scratch)

enum { TRI_RESIDUAL_ANGULAR = 0, TRI_RESIDUAL_REPROJECTION = 1 };  // TriangulationEstimator::ResidualType

struct TriRansacOptions {
  double min_tri_angle;   // radians
  double max_error;       // radians (angular residual: Create) or pixels (reprojection residual: CompleteImage)
  double confidence;      // 0.9999
  int64_t max_num_trials; // 10000
  int64_t min_num_trials; // C(n, 2) for n <= 15 (exhaustive), else 0 (CompleteImage: what the previous short track left)
  int32_t residual_type;  // TRI_RESIDUAL_*
  int32_t pad;
};

// TriangulationEstimator::Estimate for the views listed in idx[0..m): fills X, returns false when no model
__host__ __device__ inline bool tri_estimate(const TriView* views, const int* idx, int m, double min_tri_angle, double* X) {
  if (m == 2) {
    const TriView &a = views[idx[0]], &b = views[idx[1]];
    tri_two_view(a, b, X);
    return tri_positive_depth(a.P, X) && tri_positive_depth(b.P, X) && tri_angle(a.C, b.C, X) >= min_tri_angle;
  }
  double A[4][4] = {};
  for (int k = 0; k < m; ++k) tri_multi_accumulate(views[idx[k]], A);
  double v[4];
  sym4_min_eigvec(A, v);
  X[0] = v[0] / v[3]; X[1] = v[1] / v[3]; X[2] = v[2] / v[3];
  for (int k = 0; k < m; ++k)
    if (!tri_positive_depth(views[idx[k]].P, X)) return false;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j)
      if (tri_angle(views[idx[i]].C, views[idx[j]].C, X) >= min_tri_angle) return true;
  return false;
}

struct TriSupport { int num_inliers; double residual_sum; };
__host__ __device__ inline bool tri_better(const TriSupport& l, const TriSupport& r) {
  return l.num_inliers > r.num_inliers || (l.num_inliers == r.num_inliers && l.residual_sum < r.residual_sum);
}
// TriangulationEstimator::Residuals for one view: squared angular error (radians^2) or squared reprojection error (px^2)
__host__ __device__ inline double tri_residual(const TriView& v, const double* X, int residual_type) {
  if (residual_type == TRI_RESIDUAL_REPROJECTION) return tri_sq_reproj_error(v.xy, X, v.P, v.K);
  const double e = tri_angular_error(v.xn, X, v.P);
  return e * e;
}
__host__ __device__ inline TriSupport tri_support(const TriView* views, int n, const double* X, double max_residual, int residual_type, double* res) {
  TriSupport s{0, 0.0};
  for (int i = 0; i < n; ++i) {
    res[i] = tri_residual(views[i], X, residual_type);
    if (res[i] <= max_residual) { s.num_inliers++; s.residual_sum += res[i]; }
  }
  return s;
}
__host__ __device__ inline int64_t tri_num_trials(int num_inliers, int n, double confidence) {
  const double ratio = (double)num_inliers / (double)n;
  const double nom = 1.0 - confidence;
  if (nom <= 0.0) return INT64_MAX;
  const double denom = 1.0 - ratio * ratio;  // kMinNumSamples = 2
  if (denom <= 0.0) return 1;
  if (denom == 1.0) return INT64_MAX;
  return (int64_t)ceil(log(nom) / log(denom) * 3.0);
}

// EstimateTriangulation over n views: LORANSAC with lexicographic pair sampling (CombinationSampler), local
// optimisation on the inlier set (up to 10 rounds while it grows), inlier-count support with the residual sum as
// tie-break.  Returns true on success with X and the inlier mask (bit i of mask[i / 64], (n + 63) / 64 words).
// Scratch: res, best_local_res [n] doubles, idx [n] ints.
__host__ __device__ inline bool tri_ransac_scratch(const TriView* views, int n, const TriRansacOptions& o, double* X, uint64_t* mask_out,
                                                   double* res, double* best_local_res, int* idx) {
  if (n < 2) return false;
  const double max_residual = o.max_error * o.max_error;
  const int rt = o.residual_type;
  TriSupport best{0, DBL_MAX};
  double bestX[3] = {0, 0, 0};
  const int64_t all_pairs = (int64_t)n * (n - 1) / 2;
  const int64_t max_trials = o.max_num_trials < all_pairs ? o.max_num_trials : all_pairs;
  int64_t dyn_max = max_trials, trials = 0;
  int a = 0, b = 1;
  bool abort = false;
  for (trials = 0; trials < max_trials; ++trials) {
    if (abort) { trials += 1; break; }
    const int pair[2] = {a, b};
    if (++b == n) { ++a; b = a + 1; }
    double Xs[3];
    if (!tri_estimate(views, pair, 2, o.min_tri_angle, Xs)) continue;  // no model: LORANSAC tests its stop rule per model only
    const TriSupport sup = tri_support(views, n, Xs, max_residual, rt, res);
    if (tri_better(sup, best)) {
      best = sup; bestX[0] = Xs[0]; bestX[1] = Xs[1]; bestX[2] = Xs[2];
      if (sup.num_inliers > 2) {
        for (int local = 0; local < 10; ++local) {
          int m = 0;
          for (int i = 0; i < n; ++i) if (res[i] <= max_residual) idx[m++] = i;
          const int prev = best.num_inliers;
          double Xl[3];
          if (tri_estimate(views, idx, m, o.min_tri_angle, Xl)) {
            const TriSupport ls = tri_support(views, n, Xl, max_residual, rt, best_local_res);
            if (tri_better(ls, best)) {
              best = ls; bestX[0] = Xl[0]; bestX[1] = Xl[1]; bestX[2] = Xl[2];
              for (int i = 0; i < n; ++i) res[i] = best_local_res[i];  // the inlier set of the next round
            }
          }
          if (best.num_inliers <= prev) break;
        }
      }
      dyn_max = tri_num_trials(best.num_inliers, n, o.confidence);
    }
    if (trials >= dyn_max && trials >= o.min_num_trials) abort = true;
  }
  if (best.num_inliers < 2) return false;
  X[0] = bestX[0]; X[1] = bestX[1]; X[2] = bestX[2];
  for (int w = 0; w < (n + 63) / 64; ++w) mask_out[w] = 0;
  for (int i = 0; i < n; ++i)
    if (tri_residual(views[i], bestX, rt) <= max_residual) mask_out[i / 64] |= (uint64_t)1 << (i % 64);
  return true;
}

// the fixed-scratch form of the GPU batch (n <= kTriMaxViews)
__host__ __device__ inline bool tri_ransac(const TriView* views, int n, const TriRansacOptions& o, double* X, uint64_t* mask_out) {
  if (n < 2 || n > kTriMaxViews) return false;
  double res[kTriMaxViews], best_local_res[kTriMaxViews];
  int idx[kTriMaxViews];
  return tri_ransac_scratch(views, n, o, X, mask_out, res, best_local_res, idx);
}

}  // namespace mpsfm
