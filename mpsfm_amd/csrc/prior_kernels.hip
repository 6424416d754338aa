// Row f3: the per-image depth-block selection of Optimizer.__build_problem
// (reference mpsfm/sfm/mapper/bundle_adjustment.py:124-161, Appendix B of SURVEY.md) and the whitened log-depth errors
// of update_truncation_multiplier (:295-333) for ALL images of a bundle in one launch:
//   bilinear sample of the validity mask and of the depth map at every keypoint that has a 3-D point
//     (PriorUtils._data_at_kps, mpsfm/sfm/scene/image/mixins/priorutils.py:49-62: torch grid_sample, bilinear,
//      zero padding, align_corners=True, keypoints scaled by camera.sx / sy),
//   the camera-frame depth of that point (Points3DUtils.project_image_3d_points -> geometry.project3D),
//   the masks (valid == 1, depth > 0, scale filter, gross-outlier test) and the loss weights
//     magnitude = d^2 / clip(var, 1e-6),  param = m sqrt(var) / d.
// The mask decisions are booleans compared exactly with the NumPy restatement (which is pinned by vectors computed by
// the reference's own PriorUtils): the interpolation below therefore repeats its arithmetic operation by operation
// with explicitly rounded multiplies and adds (no fused multiply-add contraction).
#include <string>
#include <vector>

#include "common.h"

// hipcc contracts a * b + c into one fused multiply-add by default, and the __dmul_rn / __dadd_rn wrappers of the HIP
// headers are plain operators compiled with contraction allowed (their instructions carry the `contract` flag into
// the caller).  The bit-exact validity decision (sample == 1) needs every product and sum rounded on its own, like
// NumPy / torch on the CPU do: plain operators under this pragma.
#pragma clang fp contract(off)

namespace mpsfm {

extern thread_local std::string g_err;
int staged_upload(void* dst, const void* src, size_t bytes);
int staged_drain();
static int pfail(int code, const std::string& m) { g_err = m; return code; }
#define PRI_TRY(expr)                                                                                \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return pfail(MPSFM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct GatherArgs {
  int64_t n_obs;
  const int32_t* H; const int32_t* W; const int64_t* map_off;  // per image
  const double* sx; const double* sy; const double* q; const double* t;
  const double* depth; const uint8_t* valid;                  // concatenated maps
  const int32_t* obs_img; const double* obs_xy; const double* obs_var; const int32_t* obs_pt;
  const double* pts;
  int32_t scale_filter, gross_outliers;
  double factor, mult;
  uint8_t* flags; double* d_out; double* z_out; double* mag; double* par; double* whi;
};

// pixel coordinate of grid_sample(align_corners=True) for a keypoint coordinate k scaled by s on an axis of `size`
// samples, in the operation order of the restatement (mpsfm_amd/sfm/scene/priorutils.py:bilinear_at_kps)
__device__ __forceinline__ double grid_coord(double k, double s, int size) {
  const double sm1 = (double)(size - 1);
  double v = k * s;
  v = v / sm1;
  v = v * 2.0;
  v = v - 1.0;
  v = v + 1.0;
  v = v * 0.5;
  return v * sm1;
}

template <typename T>
__device__ __forceinline__ double bilinear(const T* map, int H, int W, double x, double y) {
  const double x0f = floor(x), y0f = floor(y);
  const double wx1 = x - x0f, wy1 = y - y0f;
  const double wx0 = 1.0 - wx1, wy0 = 1.0 - wy1;
  // out-of-range coordinates (also NaN / huge) contribute nothing: zero padding
  const bool fin = (x0f > -2.0) && (x0f < (double)W + 1.0) && (y0f > -2.0) && (y0f < (double)H + 1.0);
  if (!fin) return 0.0;
  const int x0 = (int)x0f, y0 = (int)y0f;
  double out = 0.0;
  const double w[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
    if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
      const double term = w[k] * (double)map[(size_t)yi * W + xi];
      out = out + term;
    }
  }
  return out;
}

__global__ __launch_bounds__(256) void k_depth_blocks(GatherArgs G) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G.n_obs) return;
  const int im = G.obs_img[i];
  const int H = G.H[im], W = G.W[im];
  const int64_t off = G.map_off[im];
  const double x = grid_coord(G.obs_xy[2 * i], G.sx[im], W), y = grid_coord(G.obs_xy[2 * i + 1], G.sy[im], H);
  const double v = bilinear(G.valid + off, H, W, x, y);
  const double d = bilinear(G.depth + off, H, W, x, y);
  double R[9];
  quat_to_R(G.q + 4 * im, R);
  const double* X = G.pts + 3 * (size_t)G.obs_pt[i];
  const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + G.t[3 * im + 2];
  const double var = G.obs_var[i];
  uint8_t f = 0;
  if (v == 1.0) f |= 1;
  if (d > 0.0) f |= 2;
  const double div = d / z;
  if (div < G.factor && div > 1.0 / G.factor) f |= 4;
  const double sd = sqrt(var);
  // gross-outlier test of :145-147 (log clipped from below at 1e-6, as the reference writes it)
  const double wh_g = fabs(fmax(log(d), 1e-6) - fmax(log(z), 1e-6)) / sd;
  if (wh_g < 3.0) f |= 8;
  G.flags[i] = f;
  G.d_out[i] = d;
  G.z_out[i] = z;
  G.mag[i] = d * d * (1.0 / fmax(var, 1e-6));
  G.par[i] = G.mult * sd / d;
  // whitened log-depth error of update_truncation_multiplier (:323-329)
  G.whi[i] = (log(d) - log(z)) / fmax(sd / d, 1e-6);
}

}  // namespace mpsfm

using namespace mpsfm;

extern "C" int mpsfm_depth_blocks(const mpsfm_depth_gather* g, int32_t device, uint8_t* flags, double* depth, double* depth3d,
                                  double* magnitude, double* param, double* whitened) {
  if (!g) return pfail(MPSFM_EINVAL, "gather descriptor is NULL");
  if (g->n_images < 0 || g->n_obs < 0 || g->n_pts < 0) return pfail(MPSFM_EINVAL, "negative size");
  if (g->n_obs == 0) return 0;
  if (!flags || !depth || !depth3d || !magnitude || !param || !whitened) return pfail(MPSFM_EINVAL, "output pointer is NULL");
  if (!g->map_h || !g->map_w || !g->depth_map || !g->valid_map || !g->sx || !g->sy || !g->cam_quat_xyzw || !g->cam_t)
    return pfail(MPSFM_EINVAL, "image arrays are NULL");
  if (!g->obs_img || !g->obs_xy || !g->obs_var || !g->obs_pt || !g->pts) return pfail(MPSFM_EINVAL, "observation arrays are NULL");
  if (!(g->scale_filter_factor > 0.0)) return pfail(MPSFM_EINVAL, "scale_filter_factor must be positive");
  std::vector<int64_t> off((size_t)g->n_images + 1, 0);
  for (int i = 0; i < g->n_images; ++i) {
    if (g->map_h[i] < 2 || g->map_w[i] < 2 || !g->depth_map[i] || !g->valid_map[i]) return pfail(MPSFM_EINVAL, "map missing or smaller than 2x2");
    off[(size_t)i + 1] = off[(size_t)i] + (int64_t)g->map_h[i] * g->map_w[i];
  }
  for (int64_t i = 0; i < g->n_obs; ++i)
    if (g->obs_img[i] < 0 || g->obs_img[i] >= g->n_images || g->obs_pt[i] < 0 || g->obs_pt[i] >= g->n_pts)
      return pfail(MPSFM_EINVAL, "observation index out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return pfail(MPSFM_ENODEVICE, "no HIP device visible: libmpsfm_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return pfail(MPSFM_EINVAL, "device ordinal out of range");
  if (device >= kMaxDevices) return pfail(MPSFM_EUNSUPPORTED, "device ordinals beyond 15 are not supported (per-device pools)");
  PRI_TRY(hipSetDevice(device));
  struct Blocks {  // a pooled non-blocking stream per call, never the legacy null stream (see DevBuf in tri_kernels.hip)
    std::vector<void*> v;
    hipStream_t st = nullptr;
    ~Blocks() {
      if (st) (void)hipStreamSynchronize(st);
      for (void* p : v) cached_free(p);
      release_stream(st);
    }
    void* get(size_t bytes) { void* p = cached_malloc(bytes ? bytes : 1); if (p) v.push_back(p); return p; }
  } B;
  PRI_TRY(pooled_stream(&B.st));
  const size_t ni = (size_t)g->n_images, no = (size_t)g->n_obs, npix = (size_t)off[ni];
  GatherArgs A{};
  A.n_obs = g->n_obs;
#define PRI_UP(field, T, src, count)                                                      \
  {                                                                                       \
    T* p_ = (T*)B.get(sizeof(T) * (count));                                               \
    if (!p_) return pfail(MPSFM_ENOMEM, "hipMalloc failed");                              \
    if (int rc_ = staged_upload(p_, (src), sizeof(T) * (count))) return rc_;              \
    A.field = p_;                                                                         \
  }
  PRI_UP(H, int32_t, g->map_h, ni) PRI_UP(W, int32_t, g->map_w, ni) PRI_UP(map_off, int64_t, off.data(), ni)
  PRI_UP(sx, double, g->sx, ni) PRI_UP(sy, double, g->sy, ni) PRI_UP(q, double, g->cam_quat_xyzw, 4 * ni) PRI_UP(t, double, g->cam_t, 3 * ni)
  PRI_UP(obs_img, int32_t, g->obs_img, no) PRI_UP(obs_xy, double, g->obs_xy, 2 * no) PRI_UP(obs_var, double, g->obs_var, no)
  PRI_UP(obs_pt, int32_t, g->obs_pt, no) PRI_UP(pts, double, g->pts, 3 * (size_t)g->n_pts)
#undef PRI_UP
  double* d_depth = (double*)B.get(sizeof(double) * npix);
  uint8_t* d_valid = (uint8_t*)B.get(npix);
  if (!d_depth || !d_valid) return pfail(MPSFM_ENOMEM, "hipMalloc failed");
  for (size_t i = 0; i < ni; ++i) {
    const size_t n = (size_t)(off[i + 1] - off[i]);
    if (int rc = staged_upload(d_depth + off[i], g->depth_map[i], sizeof(double) * n)) return rc;
    if (int rc = staged_upload(d_valid + off[i], g->valid_map[i], n)) return rc;
  }
  A.depth = d_depth; A.valid = d_valid;
  A.scale_filter = g->scale_filter; A.gross_outliers = g->gross_outliers; A.factor = g->scale_filter_factor; A.mult = g->multiplier;
  A.flags = (uint8_t*)B.get(no);
  double* outs[5];
  for (auto& o : outs) { o = (double*)B.get(sizeof(double) * no); if (!o) return pfail(MPSFM_ENOMEM, "hipMalloc failed"); }
  if (!A.flags) return pfail(MPSFM_ENOMEM, "hipMalloc failed");
  A.d_out = outs[0]; A.z_out = outs[1]; A.mag = outs[2]; A.par = outs[3]; A.whi = outs[4];
  if (int rc = staged_drain()) return rc;
  hipLaunchKernelGGL(k_depth_blocks, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, B.st, A);
  PRI_TRY(hipGetLastError());
  PRI_TRY(hipMemcpyAsync(flags, A.flags, no, hipMemcpyDeviceToHost, B.st));
  double* hosts[5] = {depth, depth3d, magnitude, param, whitened};
  for (int k = 0; k < 5; ++k) PRI_TRY(hipMemcpyAsync(hosts[k], outs[k], sizeof(double) * no, hipMemcpyDeviceToHost, B.st));
  PRI_TRY(hipStreamSynchronize(B.st));
  return 0;
}
