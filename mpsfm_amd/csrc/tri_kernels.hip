// Per-landmark covariance and per-track triangulation numerics (gfx950).
//   mpsfm_point_covs          replaces pycolmap.estimate_ba_covariance(POINTS), reference
//                             mpsfm/sfm/mapper/bundle_adjustment.py:244-261
//   mpsfm_triangulate_tracks  the arithmetic of COLMAP TriangulateMultiViewPoint reached through
//                             pycolmap.IncrementalTriangulator (mpsfm/sfm/mapper/triangulator.py:48,123)
//   mpsfm_filter_tracks       max pairwise triangulation angle / squared reprojection error /
//                             cheirality used by ObservationManager filters
//                             (mpsfm/sfm/mapper/base.py:686-797, reconstruction/mixins/points3D_utils.py:64-71,
//                             mpsfm/utils/geometry.py:54-75)
#include <string>
#include <vector>

#include "common.h"
#include "tri_math.h"

namespace mpsfm {

extern thread_local std::string g_err;
static int tfail(int code, const std::string& m) { g_err = m; return code; }
#define TRI_TRY(expr)                                                                                \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return tfail(MPSFM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct DevBuf {
  std::vector<void*> ptrs;
  // Every call works on a non-blocking stream of its own from the pool, never on the legacy null stream: the library is
  // called from several host threads (tests, threaded callers), and null-stream copies / fills issued by one thread while
  // another thread's solve had a deep queue of launches in flight corrupted that solve (MI355X, ROCm 7.2: reproduced with
  // tests/test_gpu_fuzz.py::test_random_problems_concurrently until the last null-stream call was gone).
  hipStream_t st = nullptr;
  DevBuf() { if (pooled_stream(&st) != hipSuccess) st = nullptr; }
  // the stream must be idle before the blocks go back to the caching allocator (an early return on a failed call would
  // otherwise free memory that is still in use) and before the stream goes back to the pool
  ~DevBuf() {
    if (st) (void)hipStreamSynchronize(st);
    for (void* p : ptrs) cached_free(p);
    release_stream(st);
  }
  // device -> caller memory, complete on return
  hipError_t down(void* host, const void* dev, size_t bytes) {
    hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  }
  template <typename T>
  T* up(const T* host, size_t n, bool copy = true) {
    void* p = nullptr;
    p = cached_malloc(std::max<size_t>(n, 1) * sizeof(T));
    if (!p) return nullptr;
    ptrs.push_back(p);
    if (copy && n && host) (void)hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, st);
    return (T*)p;
  }
};

static int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return tfail(MPSFM_ENODEVICE, "no HIP device visible: libmpsfm_hip has no CPU fallback");
  if (device < 0 || device >= n) return tfail(MPSFM_EINVAL, "device ordinal out of range");
  if (device >= kMaxDevices) return tfail(MPSFM_EUNSUPPORTED, "device ordinals beyond 15 are not supported (per-device pools)");
  TRI_TRY(hipSetDevice(device));
  return 0;
}

// ---- point covariances -------------------------------------------------------------------------
__global__ void k_pcov_accum(int64_t nobs, const int32_t* cam, const int32_t* pt, const double* q, const double* t,
                             const double* intr, const int32_t* intr_idx, const double* pts, double mag, double* H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nobs) return;
  const int c = cam[i], p = pt[i];
  double R[9];
  quat_to_R(q + 4 * c, R);
  const double* K = intr + 4 * intr_idx[c];
  const double X0 = pts[3 * p], X1 = pts[3 * p + 1], X2 = pts[3 * p + 2];
  const double Xc = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[3 * c];
  const double Yc = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[3 * c + 1];
  const double Zc = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[3 * c + 2];
  const double iz = 1.0 / Zc;
  const double a00 = K[0] * iz, a02 = -K[0] * Xc * iz * iz, a11 = K[1] * iz, a12 = -K[1] * Yc * iz * iz;
  const double j0[3] = {a00 * R[0] + a02 * R[6], a00 * R[1] + a02 * R[7], a00 * R[2] + a02 * R[8]};
  const double j1[3] = {a11 * R[3] + a12 * R[6], a11 * R[4] + a12 * R[7], a11 * R[5] + a12 * R[8]};
  double* h = H + 6 * (size_t)p;
  atomicAdd(&h[0], mag * (j0[0] * j0[0] + j1[0] * j1[0]));
  atomicAdd(&h[1], mag * (j0[0] * j0[1] + j1[0] * j1[1]));
  atomicAdd(&h[2], mag * (j0[0] * j0[2] + j1[0] * j1[2]));
  atomicAdd(&h[3], mag * (j0[1] * j0[1] + j1[1] * j1[1]));
  atomicAdd(&h[4], mag * (j0[1] * j0[2] + j1[1] * j1[2]));
  atomicAdd(&h[5], mag * (j0[2] * j0[2] + j1[2] * j1[2]));
}
__global__ void k_pcov_invert(int np, const double* H, double* cov) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  double V[6], Vi[6];
  for (int k = 0; k < 6; ++k) V[k] = H[6 * (size_t)p + k];
  double* o = cov + 9 * (size_t)p;
  if (!spd3_inverse(V, Vi)) {
    for (int k = 0; k < 9; ++k) o[k] = __builtin_nan("");
    return;
  }
  o[0] = Vi[0]; o[1] = Vi[1]; o[2] = Vi[2];
  o[3] = Vi[1]; o[4] = Vi[3]; o[5] = Vi[4];
  o[6] = Vi[2]; o[7] = Vi[4]; o[8] = Vi[5];
}

// ---- triangulation -------------------------------------------------------------------------------
struct TrackArgs {
  int32_t n_tracks;
  const double* q; const double* t; const double* intr; const int32_t* intr_idx;
  const int64_t* start; const int32_t* el_cam; const double* el_xy;
};

__global__ void k_triangulate(TrackArgs T, double* xyz) {
  const int tr = blockIdx.x * blockDim.x + threadIdx.x;
  if (tr >= T.n_tracks) return;
  double A[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int64_t e = T.start[tr]; e < T.start[tr + 1]; ++e) {
    const int cam = T.el_cam[e];
    const double* K = T.intr + 4 * T.intr_idx[cam];
    double R[9];
    quat_to_R(T.q + 4 * cam, R);
    double P[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) { P[i][0] = R[3 * i]; P[i][1] = R[3 * i + 1]; P[i][2] = R[3 * i + 2]; P[i][3] = T.t[3 * cam + i]; }
    double x[3] = {(T.el_xy[2 * e] - K[2]) / K[0], (T.el_xy[2 * e + 1] - K[3]) / K[1], 1.0};
    const double nrm = sqrt(x[0] * x[0] + x[1] * x[1] + 1.0);
    x[0] /= nrm; x[1] /= nrm; x[2] /= nrm;
    double xtP[4], term[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xtP[j] = x[0] * P[0][j] + x[1] * P[1][j] + x[2] * P[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) term[i][j] = P[i][j] - x[i] * xtP[j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) A[i][j] += term[0][i] * term[0][j] + term[1][i] * term[1][j] + term[2][i] * term[2][j];
  }
  double v[4];
  sym4_min_eigvec(A, v);
  xyz[3 * (size_t)tr] = v[0] / v[3]; xyz[3 * (size_t)tr + 1] = v[1] / v[3]; xyz[3 * (size_t)tr + 2] = v[2] / v[3];
}

__device__ inline void proj_center(const double* R, const double* t, double* C) {
  C[0] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);
  C[1] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
  C[2] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
}

__global__ void k_filter(TrackArgs T, const double* xyz, double* max_angle, double* sq_err, uint8_t* front) {
  const int tr = blockIdx.x * blockDim.x + threadIdx.x;
  if (tr >= T.n_tracks) return;
  const double X[3] = {xyz[3 * (size_t)tr], xyz[3 * (size_t)tr + 1], xyz[3 * (size_t)tr + 2]};
  const int64_t e0 = T.start[tr], e1 = T.start[tr + 1];
  double best = 0.0;
  for (int64_t e = e0; e < e1; ++e) {
    const int cam = T.el_cam[e];
    const double* K = T.intr + 4 * T.intr_idx[cam];
    double R[9];
    quat_to_R(T.q + 4 * cam, R);
    const double* tt = T.t + 3 * cam;
    const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + tt[0];
    const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + tt[1];
    const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + tt[2];
    if (front) front[e] = zc >= 2.220446049250313e-16 ? 1 : 0;
    if (sq_err) {
      const double du = K[0] * xc / zc + K[2] - T.el_xy[2 * e];
      const double dv = K[1] * yc / zc + K[3] - T.el_xy[2 * e + 1];
      sq_err[e] = du * du + dv * dv;
    }
    if (max_angle) {
      double C1[3];
      proj_center(R, tt, C1);
      for (int64_t f = e + 1; f < e1; ++f) {
        const int cam2 = T.el_cam[f];
        double R2[9], C2[3];
        quat_to_R(T.q + 4 * cam2, R2);
        proj_center(R2, T.t + 3 * cam2, C2);
        double b2 = 0, r1 = 0, r2 = 0;
#pragma unroll
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
          ang = fmin(ang, M_PI - ang);
        }
        best = fmax(best, ang);
      }
    }
  }
  if (max_angle) max_angle[tr] = best;
}

static int check_tracks(const mpsfm_tracks* T) {
  if (!T || T->n_tracks < 0 || T->n_cams < 0) return tfail(MPSFM_EINVAL, "tracks is NULL or has negative sizes");
  if (T->n_tracks > 0 && !T->track_start) return tfail(MPSFM_EINVAL, "track_start is NULL");
  if (T->n_tracks == 0) return 0;
  if (T->n_intr < 0 || (T->n_cams > 0 && (!T->cam_quat_xyzw || !T->cam_t || !T->cam_intr || !T->cam_intr_idx)))
    return tfail(MPSFM_EINVAL, "camera arrays are NULL");
  if (T->track_start[0] != 0) return tfail(MPSFM_EINVAL, "track_start[0] must be 0");
  for (int i = 0; i < T->n_tracks; ++i)
    if (T->track_start[i + 1] < T->track_start[i]) return tfail(MPSFM_EINVAL, "track_start must be non-decreasing");
  const int64_t ne = T->track_start[T->n_tracks];
  if (ne > 0 && (!T->el_cam || !T->el_xy)) return tfail(MPSFM_EINVAL, "track element arrays are NULL");
  for (int64_t e = 0; e < ne; ++e)
    if (T->el_cam[e] < 0 || T->el_cam[e] >= T->n_cams) return tfail(MPSFM_EINVAL, "el_cam out of range");
  for (int i = 0; i < T->n_cams; ++i)
    if (T->cam_intr_idx[i] < 0 || T->cam_intr_idx[i] >= T->n_intr) return tfail(MPSFM_EINVAL, "cam_intr_idx out of range");
  return 0;
}

static int upload_tracks(const mpsfm_tracks* T, DevBuf& B, TrackArgs& a) {
  const int64_t ne = T->n_tracks > 0 ? T->track_start[T->n_tracks] : 0;
  a.n_tracks = T->n_tracks;
  a.q = B.up(T->cam_quat_xyzw, (size_t)T->n_cams * 4);
  a.t = B.up(T->cam_t, (size_t)T->n_cams * 3);
  a.intr = B.up(T->cam_intr, (size_t)T->n_intr * 4);
  a.intr_idx = B.up(T->cam_intr_idx, (size_t)T->n_cams);
  a.start = B.up(T->track_start, (size_t)T->n_tracks + 1);
  a.el_cam = B.up(T->el_cam, (size_t)ne);
  a.el_xy = B.up(T->el_xy, (size_t)ne * 2);
  if (!a.q || !a.t || !a.intr || !a.intr_idx || !a.start || !a.el_cam || !a.el_xy) return tfail(MPSFM_ENOMEM, "hipMalloc failed");
  return 0;
}

}  // namespace mpsfm

using namespace mpsfm;

extern "C" {

int mpsfm_point_covs(const mpsfm_ba_problem* P, const mpsfm_ba_state* st, int32_t device, double* covs) {
  if (!P || !st || !covs) return tfail(MPSFM_EINVAL, "NULL argument");
  if (P->n_pts < 0 || P->n_cams < 0 || P->n_obs < 0 || P->n_intr < 0) return tfail(MPSFM_EINVAL, "negative size");
  if (P->n_obs > 0 && (!P->obs_cam || !P->obs_pt || !P->obs_xy)) return tfail(MPSFM_EINVAL, "observation arrays are NULL");
  if (P->n_cams > 0 && (!P->cam_intr_idx || !P->cam_intr || !st->cam_quat_xyzw || !st->cam_t)) return tfail(MPSFM_EINVAL, "camera arrays are NULL");
  if (P->n_pts > 0 && !st->pts) return tfail(MPSFM_EINVAL, "pts is NULL");
  for (int64_t i = 0; i < P->n_obs; ++i)
    if (P->obs_cam[i] < 0 || P->obs_cam[i] >= P->n_cams || P->obs_pt[i] < 0 || P->obs_pt[i] >= P->n_pts)
      return tfail(MPSFM_EINVAL, "observation index out of range");
  for (int i = 0; i < P->n_cams; ++i)
    if (P->cam_intr_idx[i] < 0 || P->cam_intr_idx[i] >= P->n_intr) return tfail(MPSFM_EINVAL, "cam_intr_idx out of range");
  if (int rc = check_device(device)) return rc;
  if (P->n_pts == 0) return 0;
  DevBuf B;
  if (!B.st) return tfail(MPSFM_EHIP, "hipStreamCreate failed");
  const int32_t* cam = B.up(P->obs_cam, (size_t)P->n_obs);
  const int32_t* pt = B.up(P->obs_pt, (size_t)P->n_obs);
  const double* q = B.up(st->cam_quat_xyzw, (size_t)P->n_cams * 4);
  const double* t = B.up(st->cam_t, (size_t)P->n_cams * 3);
  const double* intr = B.up(P->cam_intr, (size_t)P->n_intr * 4);
  const int32_t* iidx = B.up(P->cam_intr_idx, (size_t)P->n_cams);
  const double* pts = B.up(st->pts, (size_t)P->n_pts * 3);
  double* H = B.up<double>(nullptr, (size_t)P->n_pts * 6, false);
  double* dcov = B.up<double>(nullptr, (size_t)P->n_pts * 9, false);
  if (!cam || !pt || !q || !t || !intr || !iidx || !pts || !H || !dcov) return tfail(MPSFM_ENOMEM, "hipMalloc failed");
  TRI_TRY(hipMemsetAsync(H, 0, sizeof(double) * 6 * (size_t)P->n_pts, B.st));
  if (P->n_obs > 0)
    hipLaunchKernelGGL(k_pcov_accum, dim3((unsigned)((P->n_obs + 255) / 256)), dim3(256), 0, B.st, P->n_obs, cam, pt, q, t, intr,
                       iidx, pts, P->reproj_loss_magnitude, H);
  hipLaunchKernelGGL(k_pcov_invert, dim3((P->n_pts + 255) / 256), dim3(256), 0, B.st, P->n_pts, H, dcov);
  TRI_TRY(hipGetLastError());
  TRI_TRY(B.down(covs, dcov, sizeof(double) * 9 * (size_t)P->n_pts));
  return 0;
}

int mpsfm_triangulate_tracks(const mpsfm_tracks* T, int32_t device, double* xyz) {
  if (int rc = check_tracks(T)) return rc;
  if (!xyz && T->n_tracks > 0) return tfail(MPSFM_EINVAL, "xyz is NULL");
  if (int rc = check_device(device)) return rc;
  if (T->n_tracks == 0) return 0;
  DevBuf B;
  if (!B.st) return tfail(MPSFM_EHIP, "hipStreamCreate failed");
  TrackArgs a{};
  if (int rc = upload_tracks(T, B, a)) return rc;
  double* dxyz = B.up<double>(nullptr, (size_t)T->n_tracks * 3, false);
  if (!dxyz) return tfail(MPSFM_ENOMEM, "hipMalloc failed");
  hipLaunchKernelGGL(k_triangulate, dim3((T->n_tracks + 127) / 128), dim3(128), 0, B.st, a, dxyz);
  TRI_TRY(hipGetLastError());
  TRI_TRY(B.down(xyz, dxyz, sizeof(double) * 3 * (size_t)T->n_tracks));
  return 0;
}

int mpsfm_filter_tracks(const mpsfm_tracks* T, const double* xyz, int32_t device, double* max_tri_angle, double* el_sq_err,
                        uint8_t* el_front) {
  if (int rc = check_tracks(T)) return rc;
  if (!xyz && T->n_tracks > 0) return tfail(MPSFM_EINVAL, "xyz is NULL");
  if (int rc = check_device(device)) return rc;
  if (T->n_tracks == 0) return 0;
  const int64_t ne = T->track_start[T->n_tracks];
  DevBuf B;
  if (!B.st) return tfail(MPSFM_EHIP, "hipStreamCreate failed");
  TrackArgs a{};
  if (int rc = upload_tracks(T, B, a)) return rc;
  const double* dxyz = B.up(xyz, (size_t)T->n_tracks * 3);
  double* dang = max_tri_angle ? B.up<double>(nullptr, (size_t)T->n_tracks, false) : nullptr;
  double* derr = el_sq_err ? B.up<double>(nullptr, (size_t)ne, false) : nullptr;
  uint8_t* dfr = el_front ? B.up<uint8_t>(nullptr, (size_t)ne, false) : nullptr;
  if (!dxyz || (max_tri_angle && !dang) || (el_sq_err && !derr) || (el_front && !dfr)) return tfail(MPSFM_ENOMEM, "hipMalloc failed");
  hipLaunchKernelGGL(k_filter, dim3((T->n_tracks + 127) / 128), dim3(128), 0, B.st, a, dxyz, dang, derr, dfr);
  TRI_TRY(hipGetLastError());
  if (dang) TRI_TRY(B.down(max_tri_angle, dang, sizeof(double) * (size_t)T->n_tracks));
  if (derr) TRI_TRY(B.down(el_sq_err, derr, sizeof(double) * (size_t)ne));
  if (dfr) TRI_TRY(B.down(el_front, dfr, (size_t)ne));
  return 0;
}

}  // extern "C"
