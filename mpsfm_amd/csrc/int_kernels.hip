// Depth-from-normals integration (bilateral normal integration with depth priors) on gfx950:
// SURVEY.md §8f row f1.  Replaces the per-image solve of reference
// mpsfm/sfm/scene/image/integration.py:383-520 (_integrate: IRLS over a 5-point SPD system on the
// H*W log-depths, Jacobi-preconditioned CG through cupy/scipy) with matrix-free stencil kernels:
// the operators A1..A4 of generate_dx_dy (:631-680) and the normal matrix of calc_Amat (:167-234) are
// never materialised as CSR; each pixel keeps its diagonal, its right and its lower coupling.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace mpsfm {

extern thread_local std::string g_err;
static int ifail(int code, const std::string& m) { g_err = m; return code; }
#define INT_TRY(expr)                                                                                \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return ifail(MPSFM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int kIT = 256;  // threads per workgroup

struct IntDev {
  int H, W, N;
  // per pixel
  double *dp, *zp, *z, *nx, *ny, *nzu, *nzv, *Nu, *Nv;   // prepared inputs
  double *wu, *wv, *w4;                                   // w4: [4][N] wu_plus, wu_minus, wv_plus, wv_minus
  double *d, *cr, *cd, *b, *minv, *spd, *spb;             // system: diagonal, right/down coupling, rhs, 1/clip(diag)
  double *r, *zz, *p0, *p1, *q;                           // CG vectors (p double-buffered)
  double *part;                                           // [grid][4] partial sums
  double *state;                                          // [8]: rho_prev, atol^2, done, iterations, alpha-denominator ...
};

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// block sum of K values per thread -> part[block][K] (K <= 6)
template <int K>
__device__ __forceinline__ void block_partials(double (&v)[K], double* part) {
  __shared__ double s[K * (kIT / 64)];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double w = wsum(v[k]);
    if ((threadIdx.x & 63) == 0) s[k * (kIT / 64) + (threadIdx.x >> 6)] = w;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const double* p = &s[threadIdx.x * (kIT / 64)];
    part[(size_t)blockIdx.x * 8 + threadIdx.x] = (p[0] + p[1]) + (p[2] + p[3]);
  }
}
// the whole workgroup sums columns c0 .. c0+NC-1 of the partials of all blocks (fixed order:
// deterministic); every thread gets the results
template <int NC>
__device__ __forceinline__ void sum_partials(const double* part, int nblocks, int c0, double (&out)[NC]) {
  __shared__ double s[NC * (kIT / 64)];
  double v[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) v[k] = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kIT) {
#pragma unroll
    for (int k = 0; k < NC; ++k) v[k] += part[(size_t)i * 8 + c0 + k];
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double w = wsum(v[k]);
    if ((threadIdx.x & 63) == 0) s[k * (kIT / 64) + (threadIdx.x >> 6)] = w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double* p = &s[k * (kIT / 64)];
    out[k] = (p[0] + p[1]) + (p[2] + p[3]);
  }
  __syncthreads();
}

struct PrepArgs {
  int H, W;
  const double *depth_prior, *depth_unc, *normals, *nvar, *depth_init;
  const uint8_t* valid;
  double fx, fy, cx, cy, large, dmult, nmult;
};

// process_depth_prior / process_normals_prior / load_depth_checkpoint / init_int_vars (nz_u, nz_v, precisions)
__global__ __launch_bounds__(kIT) void k_int_prepare(PrepArgs a, IntDev D) {
  const int p = blockIdx.x * kIT + threadIdx.x;
  if (p >= D.N) return;
  const int row = p / a.W, col = p - row * a.W;
  const double dpr = a.depth_prior[p];
  D.dp[p] = a.dmult * (1.0 / (a.depth_unc[p] + 1e-6)) * dpr * dpr;
  D.zp[p] = log(dpr);
  D.z[p] = log(a.depth_init[p]);
  const double nx = a.normals[3 * p + 1], ny = a.normals[3 * p], nz = -a.normals[3 * p + 2];
  const bool ok = a.valid[p] != 0;
  const double m = 1.0 / a.nmult;
  const double Vnx = m * (ok ? a.nvar[3 * p + 1] : a.large), Vny = m * (ok ? a.nvar[3 * p] : a.large),
               Vnz = m * (ok ? a.nvar[3 * p + 2] : a.large);
  const double uu = (double)(a.H - 1 - row) - a.cx, vv = (double)col - a.cy;
  const double base = uu * nx + vv * ny;
  const double nzu = base + a.fx * nz, nzv = base + a.fy * nz;
  const double Du = -nx / nzu, Dv = -ny / nzv;
  D.nx[p] = nx; D.ny[p] = ny; D.nzu[p] = nzu; D.nzv[p] = nzv;
  const double a1 = uu * Du + 1.0, a2 = vv * Du;
  D.Nu[p] = 1.0 / (Vnx * (a1 * a1) + Vny * (a2 * a2) + a.fx * a.fx * Vnz * Du * Du);
  const double b1 = uu * Dv, b2 = vv * Dv + 1.0;
  D.Nv[p] = 1.0 / (Vnx * (b1 * b1) + Vny * (b2 * b2) + a.fy * a.fy * Vnz * Dv * Dv);
}

__device__ __forceinline__ double sigmoid_k(double x, double k) {
  double c = -k * x;
  c = fmin(fmax(c, -709.0), 709.0);
  return 1.0 / (1.0 + exp(c));
}

// update_W (unless the cached weights are kept), calc_Wpm and the energy of calc_energy.
// partial columns: 0 normal terms, 1 depth-prior term
__global__ __launch_bounds__(kIT) void k_int_weights(IntDev D, double kk, double lambda1, int keep_w) {
  const int p = blockIdx.x * kIT + threadIdx.x;
  double e[2] = {0.0, 0.0};
  if (p < D.N) {
    const int W = D.W, H = D.H;
    const int row = p / W, col = p - row * W;
    const double z = D.z[p];
    const double a1 = (row >= 1) ? D.nzu[p] * (D.z[p - W] - z) : 0.0;      // A1 z: top neighbour minus centre
    const double a2 = (row <= H - 2) ? D.nzu[p] * (z - D.z[p + W]) : 0.0;  // A2 z
    const double a3 = (col <= W - 2) ? D.nzv[p] * (D.z[p + 1] - z) : 0.0;  // A3 z
    const double a4 = (col >= 1) ? D.nzv[p] * (z - D.z[p - 1]) : 0.0;      // A4 z
    double wu, wv;
    if (keep_w) { wu = D.wu[p]; wv = D.wv[p]; }
    else {
      wu = sigmoid_k(a2 * a2 - a1 * a1, kk);
      wv = sigmoid_k(a4 * a4 - a3 * a3, kk);
      D.wu[p] = wu; D.wv[p] = wv;
    }
    const double wup = wu * D.Nu[p], wum = (1.0 - wu) * D.Nu[p], wvp = wv * D.Nv[p], wvm = (1.0 - wv) * D.Nv[p];
    D.w4[p] = wup; D.w4[D.N + p] = wum; D.w4[2 * (size_t)D.N + p] = wvp; D.w4[3 * (size_t)D.N + p] = wvm;
    const double nx = D.nx[p], ny = D.ny[p];
    e[0] = wup * (a1 + nx) * (a1 + nx) + wum * (a2 + nx) * (a2 + nx) + wvp * (a3 + ny) * (a3 + ny) + wvm * (a4 + ny) * (a4 + ny);
    const double dz = D.zp[p] - z;
    e[1] = lambda1 * D.dp[p] * dz * dz;
  }
  block_partials<2>(e, D.part);
}

// sparse depth term of the energy (all entries, duplicates included — calc_energy :163-164)
__global__ __launch_bounds__(kIT) void k_int_sparse_energy(int n, const int32_t* ids, const double* prec, const double* sdepth,
                                                           const double* z, double lambda2, double* out) {
  __shared__ double s[kIT / 64];
  double e = 0.0;
  for (int i = threadIdx.x; i < n; i += kIT) {
    const double dz = sdepth[i] - z[ids[i]];
    e += lambda2 * prec[i] * dz * dz;
  }
  e = wsum(e);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s[0] + s[1]) + (s[2] + s[3]);
}

// calc_Amat (:167-234) and the right-hand side (:450-459) in stencil form
__global__ __launch_bounds__(kIT) void k_int_system(IntDev D, double lambda1) {
  const int p = blockIdx.x * kIT + threadIdx.x;
  if (p >= D.N) return;
  const int W = D.W, H = D.H, N = D.N;
  const int row = p / W, col = p - row * W;
  const double* wup = D.w4; const double* wum = D.w4 + N; const double* wvp = D.w4 + 2 * (size_t)N; const double* wvm = D.w4 + 3 * (size_t)N;
  auto ttop = [&](int q) { return wup[q] * D.nzu[q] * D.nzu[q]; };
  auto tbot = [&](int q) { return wum[q] * D.nzu[q] * D.nzu[q]; };
  auto tlef = [&](int q) { return wvm[q] * D.nzv[q] * D.nzv[q]; };
  auto trig = [&](int q) { return wvp[q] * D.nzv[q] * D.nzv[q]; };
  double d = 0.0, cr = 0.0, cd = 0.0, b = 0.0;
  const double nx = D.nx[p], ny = D.ny[p];
  if (col >= 1) { d += tlef(p) + trig(p - 1); b += -D.nzv[p] * wvm[p] * ny - D.nzv[p - 1] * wvp[p - 1] * D.ny[p - 1]; }
  if (col <= W - 2) {
    const double tr = trig(p), tl = tlef(p + 1);
    d += tr + tl; cr = -(tr + tl);
    b += D.nzv[p] * wvp[p] * ny + D.nzv[p + 1] * wvm[p + 1] * D.ny[p + 1];
  }
  if (row >= 1) { d += ttop(p) + tbot(p - W); b += D.nzu[p] * wup[p] * nx + D.nzu[p - W] * wum[p - W] * D.nx[p - W]; }
  if (row <= H - 2) {
    const double tb = tbot(p), tt = ttop(p + W);
    d += tb + tt; cd = -(tb + tt);
    b += -D.nzu[p] * wum[p] * nx - D.nzu[p + W] * wup[p + W] * D.nx[p + W];
  }
  d += lambda1 * D.dp[p] + D.spd[p];
  b += lambda1 * D.dp[p] * D.zp[p] + D.spb[p];
  D.d[p] = d; D.cr[p] = cr; D.cd[p] = cd; D.b[p] = b;
  D.minv[p] = 1.0 / fmax(d, 1e-5);
}

__device__ __forceinline__ double stencil_apply(const IntDev& D, const double* x, int p, int row, int col) {
  double s = D.d[p] * x[p];
  if (col >= 1) s += D.cr[p - 1] * x[p - 1];
  if (col <= D.W - 2) s += D.cr[p] * x[p + 1];
  if (row >= 1) s += D.cd[p - D.W] * x[p - D.W];
  if (row <= D.H - 2) s += D.cd[p] * x[p + D.W];
  return s;
}

// r = b - A z, zz = M r; partials: 0 (r,zz), 1 (r,r), 2 (b,b)
__global__ __launch_bounds__(kIT) void k_cg_init(IntDev D) {
  const int p = blockIdx.x * kIT + threadIdx.x;
  double v[3] = {0.0, 0.0, 0.0};
  if (p < D.N) {
    const int row = p / D.W, col = p - row * D.W;
    const double r = D.b[p] - stencil_apply(D, D.z, p, row, col);
    const double zz = D.minv[p] * r;
    D.r[p] = r; D.zz[p] = zz;
    v[0] = r * zz; v[1] = r * r; v[2] = D.b[p] * D.b[p];
  }
  block_partials<3>(v, D.part);
}

// state: [0] rho_prev, [1] atol^2, [2] done, [3] iterations, [4] rho_cur (for the update kernel)
// Direction + matvec: p_new = zz + beta p_old, q = A p_new; partial 0: (p_new, q).
// Stops (and marks done) when |r| < atol, like scipy.sparse.linalg.cg's loop head.
__global__ __launch_bounds__(kIT) void k_cg_dir(IntDev D, int nblocks, int it, int first, double rtol) {
  double sums[3];
  sum_partials<3>(D.part, nblocks, 0, sums);  // (r,zz), (r,r), (b,b)
  const double rho = sums[0], rr = sums[1];
  const double atol2 = first ? rtol * rtol * sums[2] : D.state[1];
  const bool done = (D.state[2] != 0.0 && !first) || !(rr >= atol2);
  const double beta = first ? 0.0 : rho / D.state[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // publish for the update kernel and the next direction kernel
    D.state[5] = atol2; D.state[6] = done ? 1.0 : 0.0; D.state[4] = rho;
  }
  const double* pold = (it & 1) ? D.p1 : D.p0;
  double* pnew = (it & 1) ? D.p0 : D.p1;
  const int p = blockIdx.x * kIT + threadIdx.x;
  double v[1] = {0.0};
  if (!done && p < D.N) {
    const int W = D.W, H = D.H;
    const int row = p / W, col = p - row * W;
    auto pn = [&](int q) { return first ? D.zz[q] : D.zz[q] + beta * pold[q]; };
    const double pc = pn(p);
    double q = D.d[p] * pc;
    if (col >= 1) q += D.cr[p - 1] * pn(p - 1);
    if (col <= W - 2) q += D.cr[p] * pn(p + 1);
    if (row >= 1) q += D.cd[p - W] * pn(p - W);
    if (row <= H - 2) q += D.cd[p] * pn(p + W);
    pnew[p] = pc; D.q[p] = q;
    v[0] = pc * q;
  }
  // the partial slots 0..2 are still being read by slower workgroups of this launch: use slots 4..
  __syncthreads();
  {
    __shared__ double s[kIT / 64];
    const double w = wsum(v[0]);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) D.part[(size_t)blockIdx.x * 8 + 4] = (s[0] + s[1]) + (s[2] + s[3]);
  }
}

// x += alpha p, r -= alpha q, zz = M r; partials 0 (r,zz), 1 (r,r)
__global__ __launch_bounds__(kIT) void k_cg_update(IntDev D, int nblocks, int it) {
  double pq[1];
  sum_partials<1>(D.part, nblocks, 4, pq);
  const bool done = D.state[6] != 0.0;
  const double alpha = done ? 0.0 : D.state[4] / pq[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    D.state[1] = D.state[5];
    D.state[2] = D.state[6];
    if (!done) { D.state[0] = D.state[4]; D.state[3] += 1.0; }
  }
  const double* pnew = (it & 1) ? D.p0 : D.p1;
  const int p = blockIdx.x * kIT + threadIdx.x;
  double v[2] = {0.0, 0.0};
  if (p < D.N) {
    double r = D.r[p];
    if (!done) {
      D.z[p] += alpha * pnew[p];
      r -= alpha * D.q[p];
      D.r[p] = r;
      D.zz[p] = D.minv[p] * r;
    }
    v[0] = r * D.zz[p]; v[1] = r * r;
  }
  block_partials<2>(v, D.part);
}

__global__ __launch_bounds__(kIT) void k_int_exp(int N, const double* z, double* out) {
  const int p = blockIdx.x * kIT + threadIdx.x;
  if (p < N) out[p] = exp(z[p]);
}

struct IntPool {
  std::vector<void*> v;
  ~IntPool() { for (void* p : v) if (p) (void)hipFree(p); }
  template <typename T>
  T* get(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    v.push_back(p);
    return (T*)p;
  }
};

struct StreamGuard {
  hipStream_t st = nullptr;
  ~StreamGuard() { if (st) (void)hipStreamDestroy(st); }
};

// b = 1, x0 = 0: the system of IntegrationUncertainty (column sums of the inverse)
__global__ __launch_bounds__(kIT) void k_int_unit_rhs(IntDev D) {
  const int p = blockIdx.x * kIT + threadIdx.x;
  if (p < D.N) { D.b[p] = 1.0; D.z[p] = 0.0; }
}

// Host-side preparation shared by the two entry points: process_sparse_depth (+ the scale filter of
// _integrate when asked), device buffers, uploads.
struct IntSetup {
  IntPool pool;
  IntDev D{};
  int G = 0;
  std::vector<int32_t> ids;
  std::vector<double> sprec, sdep;
  int32_t* d_ids = nullptr;
  double *d_sp = nullptr, *d_out = nullptr, *d_in = nullptr;
  uint8_t* d_valid = nullptr;
};

static int int_check(const mpsfm_int_problem* P, int32_t device) {
  if (P->H < 2 || P->W < 2) return ifail(MPSFM_EINVAL, "map must be at least 2x2");
  if (!P->depth_prior || !P->depth_uncertainty || !P->valid || !P->normals || !P->normals_var || !P->depth_init)
    return ifail(MPSFM_EINVAL, "map pointers are NULL");
  if (P->n_sparse < 0 || (P->n_sparse > 0 && (!P->sparse_x || !P->sparse_y || !P->sparse_depth3d || !P->sparse_zvar)))
    return ifail(MPSFM_EINVAL, "sparse arrays are NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ifail(MPSFM_ENODEVICE, "no HIP device visible: libmpsfm_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return ifail(MPSFM_EINVAL, "device ordinal out of range");
  for (int i = 0; i < P->n_sparse; ++i)
    if (P->sparse_x[i] < 0 || P->sparse_x[i] >= P->W || P->sparse_y[i] < 0 || P->sparse_y[i] >= P->H)
      return ifail(MPSFM_EINVAL, "sparse pixel outside the map");
  return 0;
}

static int int_setup(const mpsfm_int_problem* P, bool use_sparse, bool scale_filter, hipStream_t st, IntSetup& U) {
  const int H = P->H, W = P->W, N = H * W;
  // NumPy "last write wins" for duplicate pixels in A and b
  std::vector<double> spd((size_t)N, 0.0), spb((size_t)N, 0.0);
  for (int i = 0; use_sparse && i < P->n_sparse; ++i) {
    const int id = P->sparse_y[i] * W + P->sparse_x[i];
    const double d3 = P->sparse_depth3d[i];
    if (scale_filter) {
      const double div = std::exp(std::log(d3)) / std::exp(std::log(P->depth_prior[id]));
      if (!(div < P->scale_filter_factor && div > 1.0 / P->scale_filter_factor)) continue;
    }
    U.ids.push_back(id); U.sprec.push_back((1.0 / P->sparse_zvar[i]) * d3 * d3); U.sdep.push_back(std::log(d3));
  }
  const auto& ids = U.ids;
  for (size_t i = 0; i < ids.size(); ++i) { spd[ids[i]] = P->lambda2 * U.sprec[i]; spb[ids[i]] = P->lambda2 * U.sprec[i] * U.sdep[i]; }

  IntPool& pool = U.pool;
  IntDev& D = U.D;
  const int G = U.G = (N + kIT - 1) / kIT;
  D.H = H; D.W = W; D.N = N;
  double* big = pool.get<double>((size_t)N * 28);
  double* d_in = U.d_in = pool.get<double>((size_t)N * 9);
  U.d_valid = pool.get<uint8_t>((size_t)N);
  D.part = pool.get<double>((size_t)G * 8);
  D.state = pool.get<double>(8);
  U.d_ids = pool.get<int32_t>(ids.size());
  U.d_sp = pool.get<double>(ids.size() * 2 + 1);
  U.d_out = pool.get<double>((size_t)N);
  if (!big || !d_in || !U.d_valid || !D.part || !D.state || !U.d_ids || !U.d_sp || !U.d_out) return ifail(MPSFM_ENOMEM, "hipMalloc failed");
  {
    double* c = big;
    auto take = [&](size_t k) { double* r = c; c += k * (size_t)N; return r; };
    D.dp = take(1); D.zp = take(1); D.z = take(1); D.nx = take(1); D.ny = take(1); D.nzu = take(1); D.nzv = take(1); D.Nu = take(1); D.Nv = take(1);
    D.wu = take(1); D.wv = take(1); D.w4 = take(4); D.d = take(1); D.cr = take(1); D.cd = take(1); D.b = take(1); D.minv = take(1);
    D.spd = take(1); D.spb = take(1); D.r = take(1); D.zz = take(1); D.p0 = take(1); D.p1 = take(1); D.q = take(1);
  }
  INT_TRY(hipMemcpyAsync(d_in, P->depth_prior, sizeof(double) * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(d_in + N, P->depth_uncertainty, sizeof(double) * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(d_in + 2 * (size_t)N, P->depth_init, sizeof(double) * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(d_in + 3 * (size_t)N, P->normals, sizeof(double) * 3 * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(d_in + 6 * (size_t)N, P->normals_var, sizeof(double) * 3 * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(U.d_valid, P->valid, (size_t)N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(D.spd, spd.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemcpyAsync(D.spb, spb.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
  INT_TRY(hipMemsetAsync(D.p0, 0, sizeof(double) * 2 * (size_t)N, st));
  if (!ids.empty()) {
    INT_TRY(hipMemcpyAsync(U.d_ids, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, st));
    INT_TRY(hipMemcpyAsync(U.d_sp, U.sprec.data(), sizeof(double) * ids.size(), hipMemcpyHostToDevice, st));
    INT_TRY(hipMemcpyAsync(U.d_sp + ids.size(), U.sdep.data(), sizeof(double) * ids.size(), hipMemcpyHostToDevice, st));
  }
  // spd / spb are host vectors that die with this frame: the copies above must have left them
  INT_TRY(hipStreamSynchronize(st));
  return 0;
}

static void int_launch_prepare(const mpsfm_int_problem* P, IntSetup& U, hipStream_t st) {
  const size_t N = (size_t)U.D.N;
  PrepArgs pa{P->H, P->W, U.d_in, U.d_in + N, U.d_in + 3 * N, U.d_in + 6 * N, U.d_in + 2 * N, U.d_valid,
              P->K[0], P->K[1], P->K[2], P->K[3], P->large_number, P->depth_magnitude_multiplier, P->normals_magnitude_multiplier};
  hipLaunchKernelGGL(k_int_prepare, dim3(U.G), dim3(kIT), 0, st, pa, U.D);
}

// preconditioned CG with scipy.sparse.linalg.cg semantics (x0 = D.z, M = 1/clip(diag), rtol); the
// host looks at the done flag every 16 iterations
static int int_run_cg(IntDev& D, int G, hipStream_t st, double rtol, int max_iter, int* its, bool* converged) {
  INT_TRY(hipMemsetAsync(D.state, 0, sizeof(double) * 7, st));
  hipLaunchKernelGGL(k_cg_init, dim3(G), dim3(kIT), 0, st, D);
  int k = 0;
  bool done = false;
  *its = 0;
  while (!done && k < max_iter) {
    const int batch = std::min(16, max_iter - k);
    for (int j = 0; j < batch; ++j, ++k) {
      hipLaunchKernelGGL(k_cg_dir, dim3(G), dim3(kIT), 0, st, D, G, k, k == 0 ? 1 : 0, rtol);
      hipLaunchKernelGGL(k_cg_update, dim3(G), dim3(kIT), 0, st, D, G, k);
    }
    double hs[8];
    INT_TRY(hipMemcpyAsync(hs, D.state, sizeof(hs), hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
    done = hs[2] != 0.0;
    *its = (int)hs[3];
  }
  INT_TRY(hipGetLastError());
  if (converged) *converged = done;
  return 0;
}

}  // namespace mpsfm

using namespace mpsfm;

extern "C" int mpsfm_integrate_depth(const mpsfm_int_problem* P, int32_t device, double* depth_out, mpsfm_int_summary* S) {
  if (!P || !depth_out || !S) return ifail(MPSFM_EINVAL, "NULL argument");
  if (P->max_iter < 0 || P->max_iter > MPSFM_INT_MAX_IRLS) return ifail(MPSFM_EINVAL, "max_iter out of range");
  if (int rc = int_check(P, device)) return rc;
  INT_TRY(hipSetDevice(device));
  std::memset(S, 0, sizeof(*S));
  // a stream of its own: concurrent calls from different host threads (one image each) overlap on the GPU
  StreamGuard sg;
  INT_TRY(hipStreamCreateWithFlags(&sg.st, hipStreamNonBlocking));
  hipStream_t st = sg.st;
  const int N = P->H * P->W;
  IntSetup U;
  if (int rc = int_setup(P, true, P->scale_filter != 0, st, U)) return rc;
  IntDev& D = U.D;
  const int G = U.G;
  const auto& ids = U.ids;
  int32_t* d_ids = U.d_ids;
  double* d_sp = U.d_sp;
  double* d_out = U.d_out;
  const bool keep_w = P->init && P->integrated && P->wu && P->wv;
  if (keep_w) {
    INT_TRY(hipMemcpyAsync(D.wu, P->wu, sizeof(double) * N, hipMemcpyHostToDevice, st));
    INT_TRY(hipMemcpyAsync(D.wv, P->wv, sizeof(double) * N, hipMemcpyHostToDevice, st));
  }
  hipEvent_t e0, e1;
  INT_TRY(hipEventCreate(&e0)); INT_TRY(hipEventCreate(&e1));
  INT_TRY(hipEventRecord(e0, st));
  int_launch_prepare(P, U, st);

  std::vector<double> hpart((size_t)G * 8);
  auto energy = [&](int keep, double* out) -> int {
    hipLaunchKernelGGL(k_int_weights, dim3(G), dim3(kIT), 0, st, D, P->k, P->lambda1, keep);
    hipLaunchKernelGGL(k_int_sparse_energy, dim3(1), dim3(kIT), 0, st, (int)ids.size(), d_ids, d_sp, d_sp + ids.size(), D.z, P->lambda2,
                       D.state + 7);
    INT_TRY(hipMemcpyAsync(hpart.data(), D.part, sizeof(double) * hpart.size(), hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
    double e_sparse;
    INT_TRY(hipMemcpyAsync(&e_sparse, D.state + 7, sizeof(double), hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
    double e_n = 0.0, e_d = 0.0;
    for (int i = 0; i < G; ++i) { e_n += hpart[(size_t)i * 8]; e_d += hpart[(size_t)i * 8 + 1]; }
    *out = e_n + e_d + (ids.empty() ? 0.0 : e_sparse);
    return 0;
  };
  auto finish = [&](int rc) {
    float ms = 0.f;
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    S->ms = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
  };
  auto save_weights = [&]() -> int {
    if (P->wu) INT_TRY(hipMemcpyAsync(P->wu, D.wu, sizeof(double) * N, hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
    if (P->wv) INT_TRY(hipMemcpyAsync(P->wv, D.wv, sizeof(double) * N, hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
    return 0;
  };

  double en = 0.0;
  if (int rc = energy(keep_w ? 1 : 0, &en)) return finish(rc);
  S->energy_initial = S->energy_final = en;
  S->energies[0] = en;
  S->energy_old_out = P->energy_old;
  S->integrated_out = P->integrated;
  if (P->integrated && !(std::fabs(en - P->energy_old) / P->energy_old > P->tol)) {
    if (int rc = save_weights()) return finish(rc);
    return finish(0);  // energy has not changed: skip this frame (:433-437)
  }
  const double energy_0 = en;
  double min_energy = en;
  bool success = true;
  for (int it = 0; it < P->max_iter; ++it) {
    hipLaunchKernelGGL(k_int_system, dim3(G), dim3(kIT), 0, st, D, P->lambda1);
    int cg_its = 0;
    if (int rc = int_run_cg(D, G, st, P->cg_tol, P->cg_max_iter, &cg_its, nullptr)) return finish(rc);
    S->cg_iters[it] = cg_its;
    S->cg_iterations_total += cg_its;
    const double energy_old = en;
    min_energy = std::min(en, min_energy);
    if (int rc = energy(0, &en)) return finish(rc);
    S->irls_iterations = it + 1;
    S->energies[it + 1] = en;
    const double rel = std::fabs(en - energy_old) / energy_old, rel_min = std::fabs(en - min_energy) / min_energy;
    if (((rel < P->tol && (energy_old - en) > 0) || (rel_min < P->tol && (min_energy - en) > 0)) && en < energy_0) break;
    if (en > energy_0) { success = false; break; }
  }
  S->energy_final = en;
  S->integrated_out = 1;
  if (int rc = save_weights()) return finish(rc);
  if (!success) {  // energy increased: keep the old map (:504-508)
    S->energy_old_out = energy_0;
    return finish(0);
  }
  S->energy_old_out = en;
  S->changed = 1;
  hipLaunchKernelGGL(k_int_exp, dim3(G), dim3(kIT), 0, st, N, D.z, d_out);
  INT_TRY(hipMemcpyAsync(depth_out, d_out, sizeof(double) * N, hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
  return finish(0);
}

// IntegrationUncertainty.solve (reference integration.py:51-79) on the matrix of calculate_hessian
// (:522-574).  The reference solves H x = e_k for every query pixel k and returns x.sum(0) (:77), i.e. the
// column sum of H^-1; H is symmetric, so that is (H^-1 1)[k] and ONE solve serves every query.
extern "C" int mpsfm_integration_variances(const mpsfm_int_problem* P, int32_t device, int32_t use_sparse, int32_t n_query,
                                           const int32_t* qx, const int32_t* qy, double rtol, int32_t max_iter,
                                           double* var_out, double* field_out, mpsfm_int_summary* S) {
  if (!P || !S) return ifail(MPSFM_EINVAL, "NULL argument");
  if (n_query < 0 || (n_query > 0 && (!qx || !qy || !var_out))) return ifail(MPSFM_EINVAL, "query arrays are NULL");
  if (!(rtol > 0.0) || max_iter <= 0) return ifail(MPSFM_EINVAL, "rtol / max_iter must be positive");
  if (int rc = int_check(P, device)) return rc;
  for (int i = 0; i < n_query; ++i)
    if (qx[i] < 0 || qx[i] >= P->W || qy[i] < 0 || qy[i] >= P->H) return ifail(MPSFM_EINVAL, "query pixel outside the map");
  INT_TRY(hipSetDevice(device));
  std::memset(S, 0, sizeof(*S));
  StreamGuard sg;
  INT_TRY(hipStreamCreateWithFlags(&sg.st, hipStreamNonBlocking));
  hipStream_t st = sg.st;
  const int N = P->H * P->W;
  IntSetup U;
  if (int rc = int_setup(P, use_sparse != 0, false, st, U)) return rc;  // calculate_hessian applies no scale filter (:542)
  IntDev& D = U.D;
  hipEvent_t e0, e1;
  INT_TRY(hipEventCreate(&e0)); INT_TRY(hipEventCreate(&e1));
  INT_TRY(hipEventRecord(e0, st));
  int_launch_prepare(P, U, st);
  hipLaunchKernelGGL(k_int_weights, dim3(U.G), dim3(kIT), 0, st, D, P->k, P->lambda1, 0);  // init=False: weights from the checkpoint
  hipLaunchKernelGGL(k_int_system, dim3(U.G), dim3(kIT), 0, st, D, P->lambda1);
  hipLaunchKernelGGL(k_int_unit_rhs, dim3(U.G), dim3(kIT), 0, st, D);
  int its = 0;
  bool conv = false;
  int rc = int_run_cg(D, U.G, st, rtol, max_iter, &its, &conv);
  float ms = 0.f;
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc) return rc;
  S->ms = ms;
  S->irls_iterations = 1;
  S->cg_iters[0] = its;
  S->cg_iterations_total = its;
  S->changed = conv ? 1 : 0;
  std::vector<double> host;
  double* f = field_out;
  if (!f) { host.resize((size_t)N); f = host.data(); }
  INT_TRY(hipMemcpyAsync(f, D.z, sizeof(double) * N, hipMemcpyDeviceToHost, st));
  INT_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < n_query; ++i) var_out[i] = f[(size_t)qy[i] * P->W + qx[i]];
  return 0;
}
