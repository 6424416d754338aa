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
// pinned staging uploads (ba_solver.hip): pageable caller memory is not handed to the runtime directly
int staged_upload(void* dst, const void* src, size_t bytes);
int staged_drain();
static int ifail(int code, const std::string& m) { g_err = m; return code; }
#define INT_TRY(expr)                                                                                \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return ifail(MPSFM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int kIT = 256;  // threads per workgroup
// Pixels per thread in the CG kernels, chosen per solve: one image (112 k pixels) is latency-bound and fastest
// with 1; a batch of full-size maps is bandwidth-bound and 23 % faster with 2 (more bytes in flight per lane).
constexpr size_t kPixSwitch = 400000;  // total pixels of the batch from which 2 pixels per thread are used

// A batch of B images of one size: every per-pixel array is [B][N] (w4: [B][4][N]); part is [B][G][8],
// state [B][8].  Kernels are launched on a (G, B) grid; blockIdx.y picks the image.
struct IntDev {
  int H, W, N, B, G;
  // per pixel
  double *dp, *zp, *z, *nx, *ny, *nzu, *nzv, *Nu, *Nv;   // prepared inputs
  double *wu, *wv, *w4;                                   // w4: [4][N] wu_plus, wu_minus, wv_plus, wv_minus
  double *d, *cr, *cd, *b, *minv, *spd, *spb;             // system: diagonal, right/down coupling, rhs, 1/clip(diag)
  double *r, *zz, *p0, *p1, *q;                           // CG vectors (p double-buffered)
  double *part;                                           // [grid][8] partial sums
  double *state;                                          // [8]: rho_prev, atol^2, done, iterations, alpha-denominator ...
  const int32_t* act;                                     // [B] 1: the image takes part in this launch
};

// the view of image b: all pointers advanced to its slice
__device__ __forceinline__ IntDev int_image(IntDev D, int b) {
  const size_t o = (size_t)b * (size_t)D.N;
  D.dp += o; D.zp += o; D.z += o; D.nx += o; D.ny += o; D.nzu += o; D.nzv += o; D.Nu += o; D.Nv += o;
  D.wu += o; D.wv += o; D.w4 += 4 * o;
  D.d += o; D.cr += o; D.cd += o; D.b += o; D.minv += o; D.spd += o; D.spb += o;
  D.r += o; D.zz += o; D.p0 += o; D.p1 += o; D.q += o;
  D.part += (size_t)b * (size_t)D.G * 8;
  D.state += (size_t)b * 8;
  return D;
}

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
  const double *depth_prior, *depth_unc, *normals, *nvar, *depth_init;  // [B][N] (normals, nvar: [B][N][3])
  const uint8_t* valid;                                                  // [B][N]
  const double* K;                                                       // [B][4] fx fy cx cy in map pixels
  double large, dmult, nmult;
};

// process_depth_prior / process_normals_prior / load_depth_checkpoint / init_int_vars (nz_u, nz_v, precisions)
__global__ __launch_bounds__(kIT) void k_int_prepare(PrepArgs a, IntDev Dall) {
  const int bimg = blockIdx.y;
  if (!Dall.act[bimg]) return;
  const IntDev D = int_image(Dall, bimg);
  const int p = blockIdx.x * kIT + threadIdx.x;
  if (p >= D.N) return;
  const size_t o = (size_t)bimg * (size_t)D.N;
  const double fx = a.K[4 * bimg], fy = a.K[4 * bimg + 1], cx = a.K[4 * bimg + 2], cy = a.K[4 * bimg + 3];
  const double* nrm = a.normals + 3 * o; const double* nvr = a.nvar + 3 * o;
  const int row = p / a.W, col = p - row * a.W;
  const double dpr = a.depth_prior[o + p];
  D.dp[p] = a.dmult * (1.0 / (a.depth_unc[o + p] + 1e-6)) * dpr * dpr;
  D.zp[p] = log(dpr);
  D.z[p] = log(a.depth_init[o + p]);
  const double nx = nrm[3 * p + 1], ny = nrm[3 * p], nz = -nrm[3 * p + 2];
  const bool ok = a.valid[o + p] != 0;
  const double m = 1.0 / a.nmult;
  const double Vnx = m * (ok ? nvr[3 * p + 1] : a.large), Vny = m * (ok ? nvr[3 * p] : a.large),
               Vnz = m * (ok ? nvr[3 * p + 2] : a.large);
  const double uu = (double)(a.H - 1 - row) - cx, vv = (double)col - cy;
  const double base = uu * nx + vv * ny;
  const double nzu = base + fx * nz, nzv = base + fy * nz;
  const double Du = -nx / nzu, Dv = -ny / nzv;
  D.nx[p] = nx; D.ny[p] = ny; D.nzu[p] = nzu; D.nzv[p] = nzv;
  const double a1 = uu * Du + 1.0, a2 = vv * Du;
  D.Nu[p] = 1.0 / (Vnx * (a1 * a1) + Vny * (a2 * a2) + fx * fx * Vnz * Du * Du);
  const double b1 = uu * Dv, b2 = vv * Dv + 1.0;
  D.Nv[p] = 1.0 / (Vnx * (b1 * b1) + Vny * (b2 * b2) + fy * fy * Vnz * Dv * Dv);
}

__device__ __forceinline__ double sigmoid_k(double x, double k) {
  double c = -k * x;
  c = fmin(fmax(c, -709.0), 709.0);
  return 1.0 / (1.0 + exp(c));
}

// update_W (unless the cached weights are kept), calc_Wpm and the energy of calc_energy.
// partial columns: 0 normal terms, 1 depth-prior term
__global__ __launch_bounds__(kIT) void k_int_weights(IntDev Dall, double kk, double lambda1, const int32_t* keep_w_img) {
  if (!Dall.act[blockIdx.y]) return;
  const IntDev D = int_image(Dall, blockIdx.y);
  const int keep_w = keep_w_img[blockIdx.y];
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

// sparse depth term of the energy (all entries, duplicates included — calc_energy :163-164); one
// workgroup per image, entries of image b are [off[b], off[b+1])
__global__ __launch_bounds__(kIT) void k_int_sparse_energy(IntDev Dall, const int32_t* off, const int32_t* ids, const double* prec,
                                                           const double* sdepth, double lambda2, double* out) {
  const int bimg = blockIdx.x;
  if (!Dall.act[bimg]) return;
  const double* z = Dall.z + (size_t)bimg * (size_t)Dall.N;
  __shared__ double s[kIT / 64];
  double e = 0.0;
  for (int i = off[bimg] + (int)threadIdx.x; i < off[bimg + 1]; i += kIT) {
    const double dz = sdepth[i] - z[ids[i]];
    e += lambda2 * prec[i] * dz * dz;
  }
  e = wsum(e);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) out[bimg] = (s[0] + s[1]) + (s[2] + s[3]);
}

// calc_Amat (:167-234) and the right-hand side (:450-459) in stencil form
__global__ __launch_bounds__(kIT) void k_int_system(IntDev Dall, double lambda1) {
  if (!Dall.act[blockIdx.y]) return;
  const IntDev D = int_image(Dall, blockIdx.y);
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
template <int PIX>
__global__ __launch_bounds__(kIT) void k_cg_init(IntDev Dall) {
  if (!Dall.act[blockIdx.y]) return;
  const IntDev D = int_image(Dall, blockIdx.y);
  double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < PIX; ++u) {
    const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
    if (p < D.N) {
      const int row = p / D.W, col = p - row * D.W;
      const double r = D.b[p] - stencil_apply(D, D.z, p, row, col);
      const double zz = D.minv[p] * r;
      D.r[p] = r; D.zz[p] = zz;
      v[0] += r * zz; v[1] += r * r; v[2] += D.b[p] * D.b[p];
    }
  }
  block_partials<3>(v, D.part);
}

// state: [0] rho_prev, [1] atol^2, [2] done, [3] iterations, [4] rho_cur (for the update kernel)
// Direction + matvec: p_new = zz + beta p_old, q = A p_new; partial 0: (p_new, q).
// Stops (and marks done) when |r| < atol, like scipy.sparse.linalg.cg's loop head.
template <int PIX>
__global__ __launch_bounds__(kIT) void k_cg_dir(IntDev Dall, int nblocks, int it, int first, double rtol) {
  if (!Dall.act[blockIdx.y]) return;
  const IntDev D = int_image(Dall, blockIdx.y);
  // This image's solve has met its test in an earlier iteration (the host only looks every 16): nothing left to do,
  // and in a batch a finished image stops taking bandwidth from the others.  The flag is requested now and tested
  // after the operand loads have been issued (one-pixel variant), so the test adds no memory round trip of its own.
  const double finished = first ? 0.0 : D.state[2];
  if (PIX != 1 && finished != 0.0) return;
  const double* pold = (it & 1) ? D.p1 : D.p0;
  double* pnew = (it & 1) ? D.p0 : D.p1;
  const int W = D.W, H = D.H;
  // Operands first: their loads do not depend on the scalars of the prologue below, whose chain (read the
  // partials of every workgroup, reduce, barrier) is about half of this kernel's duration.  Absent neighbours
  // get a zero coupling, so the arithmetic needs no further bounds tests.
  // (Only with one pixel per thread, the latency-bound regime: with two, the bandwidth-bound one, the extra live
  // registers cost more occupancy than the overlap gains — 48.6 vs 60.3 ms on 12 full-size maps.)
  constexpr bool kPreload = (PIX == 1);
  double zc[PIX][5], pc_[PIX][5], dd[PIX], wl[PIX], wr[PIX], wu[PIX], wd[PIX];
  auto load_operands = [&]() {
#pragma unroll
  for (int u = 0; u < PIX; ++u) {
    const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
    const bool in = p < D.N;
    const int row = in ? p / W : 0, col = in ? p - row * W : 0;
    const bool hl = in && col >= 1, hr = in && col <= W - 2, hu = in && row >= 1, hd = in && row <= H - 2;
    const int nb[5] = {p, p - 1, p + 1, p - W, p + W};
    const bool ok[5] = {in, hl, hr, hu, hd};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      zc[u][k] = ok[k] ? D.zz[nb[k]] : 0.0;
      pc_[u][k] = (ok[k] && !first) ? pold[nb[k]] : 0.0;
    }
    dd[u] = in ? D.d[p] : 0.0;
    wl[u] = hl ? D.cr[p - 1] : 0.0; wr[u] = hr ? D.cr[p] : 0.0;
    wu[u] = hu ? D.cd[p - W] : 0.0; wd[u] = hd ? D.cd[p] : 0.0;
  }
  };
  if (kPreload) load_operands();
  if (PIX == 1 && finished != 0.0) return;
  double sums[3];
  sum_partials<3>(D.part, nblocks, 0, sums);  // (r,zz), (r,r), (b,b)
  const double rho = sums[0], rr = sums[1];
  const double atol2 = first ? rtol * rtol * sums[2] : D.state[1];
  const bool done = (D.state[2] != 0.0 && !first) || !(rr >= atol2);
  const double beta = first ? 0.0 : rho / D.state[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // publish for the update kernel and the next direction kernel
    D.state[5] = atol2; D.state[6] = done ? 1.0 : 0.0; D.state[4] = rho;
  }
  double v[1] = {0.0};
  if constexpr (!kPreload) {
    // streaming form: operands are consumed as they arrive, pixel by pixel (fewer live registers)
#pragma unroll
    for (int u = 0; u < PIX; ++u) {
      const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
      if (!done && p < D.N) {
        const int row = p / W, col = p - row * W;
        auto pnv = [&](int q) { return first ? D.zz[q] : D.zz[q] + beta * pold[q]; };
        const double pc = pnv(p);
        double q = D.d[p] * pc;
        if (col >= 1) q += D.cr[p - 1] * pnv(p - 1);
        if (col <= W - 2) q += D.cr[p] * pnv(p + 1);
        if (row >= 1) q += D.cd[p - W] * pnv(p - W);
        if (row <= H - 2) q += D.cd[p] * pnv(p + W);
        pnew[p] = pc; D.q[p] = q;
        v[0] += pc * q;
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < PIX; ++u) {
    const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
    if (!done && p < D.N) {
      double pn[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) pn[k] = first ? zc[u][k] : zc[u][k] + beta * pc_[u][k];
      const double pc = pn[0];
      // same order of additions as the reference stencil: diagonal, left, right, up, down
      double q = dd[u] * pc;
      q += wl[u] * pn[1];
      q += wr[u] * pn[2];
      q += wu[u] * pn[3];
      q += wd[u] * pn[4];
      pnew[p] = pc; D.q[p] = q;
      v[0] += pc * q;
    }
  }
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
template <int PIX>
__global__ __launch_bounds__(kIT) void k_cg_update(IntDev Dall, int nblocks, int it) {
  if (!Dall.act[blockIdx.y]) return;
  const IntDev D = int_image(Dall, blockIdx.y);
  const double finished = D.state[2];  // set in an earlier iteration by this kernel's block 0 (below); tested after the loads
  if (PIX != 1 && finished != 0.0) return;
  const double* pnew = (it & 1) ? D.p0 : D.p1;
  // operands before the prologue (see k_cg_dir)
  constexpr bool kPreload = (PIX == 1);
  double lp[PIX], lq[PIX], lr[PIX], lz[PIX], lm[PIX], lzz[PIX];
  auto load_operands = [&]() {
#pragma unroll
    for (int u = 0; u < PIX; ++u) {
      const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
      const bool in = p < D.N;
      lp[u] = in ? pnew[p] : 0.0; lq[u] = in ? D.q[p] : 0.0; lr[u] = in ? D.r[p] : 0.0;
      lz[u] = in ? D.z[p] : 0.0; lm[u] = in ? D.minv[p] : 0.0; lzz[u] = in ? D.zz[p] : 0.0;
    }
  };
  if (kPreload) load_operands();
  if (PIX == 1 && finished != 0.0) return;
  double pq[1];
  sum_partials<1>(D.part, nblocks, 4, pq);
  const bool done = D.state[6] != 0.0;
  const double alpha = done ? 0.0 : D.state[4] / pq[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    D.state[1] = D.state[5];
    D.state[2] = D.state[6];
    if (!done) { D.state[0] = D.state[4]; D.state[3] += 1.0; }
  }
  double v[2] = {0.0, 0.0};
  if constexpr (!kPreload) {
#pragma unroll
    for (int u = 0; u < PIX; ++u) {
      const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
      if (p < D.N) {
        double r = D.r[p];
        if (!done) {
          D.z[p] += alpha * pnew[p];
          r -= alpha * D.q[p];
          D.r[p] = r;
          D.zz[p] = D.minv[p] * r;
        }
        v[0] += r * D.zz[p]; v[1] += r * r;
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < PIX; ++u) {
    const int p = (blockIdx.x * PIX + u) * kIT + threadIdx.x;
    if (p < D.N) {
      double r = lr[u], zz = lzz[u];
      if (!done) {
        D.z[p] = lz[u] + alpha * lp[u];
        r -= alpha * lq[u];
        zz = lm[u] * r;
        D.r[p] = r;
        D.zz[p] = zz;
      }
      v[0] += r * zz; v[1] += r * r;
    }
  }
  }
  block_partials<2>(v, D.part);
}

__global__ __launch_bounds__(kIT) void k_int_exp(size_t n, const double* z, double* out) {
  const size_t p = (size_t)blockIdx.x * kIT + threadIdx.x;
  if (p < n) out[p] = exp(z[p]);
}

struct IntPool {
  std::vector<void*> v, vh;
  ~IntPool() {
    for (void* p : v) cached_free(p);
    for (void* p : vh) if (p) (void)hipHostFree(p);
  }
  // pinned host memory: the per-iteration read-backs (done flags, energy partials) are plain DMA, not staged copies
  template <typename T>
  T* get_host(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) return nullptr;
    vh.push_back(p);
    return (T*)p;
  }
  template <typename T>
  T* get(size_t n) {
    void* p = cached_malloc(std::max<size_t>(n, 1) * sizeof(T));
    if (!p) return nullptr;
    v.push_back(p);
    return (T*)p;
  }
};

struct StreamGuard {
  hipStream_t st = nullptr;
  ~StreamGuard() {
    if (st) { (void)hipStreamSynchronize(st); release_stream(st); }  // back to the pool, idle
  }
};

// b = 1, x0 = 0: the system of IntegrationUncertainty (column sums of the inverse)
__global__ __launch_bounds__(kIT) void k_int_unit_rhs(IntDev Dall) {
  if (!Dall.act[blockIdx.y]) return;
  const IntDev D = int_image(Dall, blockIdx.y);
  const int p = blockIdx.x * kIT + threadIdx.x;
  if (p < D.N) { D.b[p] = 1.0; D.z[p] = 0.0; }
}

// ---- host side ---------------------------------------------------------------------------------------
// A batch of images of one size and one configuration, resident on the device.
struct IntBatch {
  IntPool pool;
  IntDev D{};
  int B = 0, G = 0;
  size_t N = 0;
  std::vector<int32_t> sp_off;            // [B+1] sparse entries per image (after the scale filter)
  std::vector<int32_t> ids;
  std::vector<double> sprec, sdep;
  int32_t *d_ids = nullptr, *d_off = nullptr, *d_act = nullptr, *d_keep = nullptr;
  double *d_sp = nullptr, *d_out = nullptr, *d_in = nullptr, *d_K = nullptr, *d_esp = nullptr;
  uint8_t* d_valid = nullptr;
  std::vector<int32_t> act;               // host copy of the activity mask
  double *hpart = nullptr, *hstate = nullptr, *hesp = nullptr;  // pinned: [B][G][8], [B][8], [B]
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
  if (device >= kMaxDevices) return ifail(MPSFM_EUNSUPPORTED, "device ordinals beyond 15 are not supported (per-device pools)");
  for (int i = 0; i < P->n_sparse; ++i)
    if (P->sparse_x[i] < 0 || P->sparse_x[i] >= P->W || P->sparse_y[i] < 0 || P->sparse_y[i] >= P->H)
      return ifail(MPSFM_EINVAL, "sparse pixel outside the map");
  return 0;
}

// images of a batch must agree in size and in every configuration value the kernels take as a scalar
static bool int_same_config(const mpsfm_int_problem& a, const mpsfm_int_problem& b) {
  return a.H == b.H && a.W == b.W && a.large_number == b.large_number && a.tol == b.tol && a.step_size == b.step_size &&
         a.cg_tol == b.cg_tol && a.lambda1 == b.lambda1 && a.lambda2 == b.lambda2 && a.k == b.k &&
         a.depth_magnitude_multiplier == b.depth_magnitude_multiplier && a.normals_magnitude_multiplier == b.normals_magnitude_multiplier &&
         a.scale_filter_factor == b.scale_filter_factor && a.max_iter == b.max_iter && a.cg_max_iter == b.cg_max_iter &&
         a.scale_filter == b.scale_filter;
}

// process_sparse_depth (+ the scale filter of _integrate when asked), device buffers, uploads
static int int_setup(const mpsfm_int_problem* Ps, int B, bool use_sparse, bool scale_filter, hipStream_t st, IntBatch& U) {
  const mpsfm_int_problem& P0 = Ps[0];
  const int H = P0.H, W = P0.W;
  const size_t N = (size_t)H * W;
  U.B = B; U.N = N;
  // NumPy "last write wins" for duplicate pixels in A and b
  std::vector<double> spd((size_t)B * N, 0.0), spb((size_t)B * N, 0.0), Kh((size_t)B * 4);
  U.sp_off.assign((size_t)B + 1, 0);
  for (int b = 0; b < B; ++b) {
    const mpsfm_int_problem* P = &Ps[b];
    const size_t first = U.ids.size();
    for (int i = 0; use_sparse && i < P->n_sparse; ++i) {
      const int id = P->sparse_y[i] * W + P->sparse_x[i];
      const double d3 = P->sparse_depth3d[i];
      if (scale_filter) {
        const double div = std::exp(std::log(d3)) / std::exp(std::log(P->depth_prior[id]));
        if (!(div < P->scale_filter_factor && div > 1.0 / P->scale_filter_factor)) continue;
      }
      U.ids.push_back(id); U.sprec.push_back((1.0 / P->sparse_zvar[i]) * d3 * d3); U.sdep.push_back(std::log(d3));
    }
    for (size_t i = first; i < U.ids.size(); ++i) {
      spd[(size_t)b * N + U.ids[i]] = P->lambda2 * U.sprec[i];
      spb[(size_t)b * N + U.ids[i]] = P->lambda2 * U.sprec[i] * U.sdep[i];
    }
    U.sp_off[(size_t)b + 1] = (int32_t)U.ids.size();
    for (int k = 0; k < 4; ++k) Kh[(size_t)b * 4 + k] = P->K[k];
  }
  const auto& ids = U.ids;
  IntPool& pool = U.pool;
  IntDev& D = U.D;
  const int G = U.G = (int)((N + kIT - 1) / kIT);
  D.H = H; D.W = W; D.N = (int)N; D.B = B; D.G = G;
  const size_t BN = (size_t)B * N;
  double* big = pool.get<double>(BN * 28);
  double* d_in = U.d_in = pool.get<double>(BN * 9);
  U.d_valid = pool.get<uint8_t>(BN);
  D.part = pool.get<double>((size_t)B * G * 8);
  D.state = pool.get<double>((size_t)B * 8);
  U.d_ids = pool.get<int32_t>(ids.size());
  U.d_off = pool.get<int32_t>((size_t)B + 1);
  U.d_act = pool.get<int32_t>((size_t)B);
  U.d_keep = pool.get<int32_t>((size_t)B);
  U.d_sp = pool.get<double>(ids.size() * 2 + 1);
  U.d_out = pool.get<double>(BN);
  U.d_K = pool.get<double>((size_t)B * 4);
  U.d_esp = pool.get<double>((size_t)B);
  if (!big || !d_in || !U.d_valid || !D.part || !D.state || !U.d_ids || !U.d_off || !U.d_act || !U.d_keep || !U.d_sp || !U.d_out || !U.d_K || !U.d_esp)
    return ifail(MPSFM_ENOMEM, "hipMalloc failed");
  D.act = U.d_act;
  {
    double* c = big;
    auto take = [&](size_t k) { double* r = c; c += k * BN; return r; };
    D.dp = take(1); D.zp = take(1); D.z = take(1); D.nx = take(1); D.ny = take(1); D.nzu = take(1); D.nzv = take(1); D.Nu = take(1); D.Nv = take(1);
    D.wu = take(1); D.wv = take(1); D.w4 = take(4); D.d = take(1); D.cr = take(1); D.cd = take(1); D.b = take(1); D.minv = take(1);
    D.spd = take(1); D.spb = take(1); D.r = take(1); D.zz = take(1); D.p0 = take(1); D.p1 = take(1); D.q = take(1);
  }
  // inputs: [prior | unc | init] each [B][N], then normals [B][N][3], nvar [B][N][3]
  auto up = [&](void* dst, const void* src, size_t bytes) { return staged_upload(dst, src, bytes); };
  for (int b = 0; b < B; ++b) {
    const mpsfm_int_problem* P = &Ps[b];
    if (int rc = up(d_in + (size_t)b * N, P->depth_prior, sizeof(double) * N)) return rc;
    if (int rc = up(d_in + BN + (size_t)b * N, P->depth_uncertainty, sizeof(double) * N)) return rc;
    if (int rc = up(d_in + 2 * BN + (size_t)b * N, P->depth_init, sizeof(double) * N)) return rc;
    if (int rc = up(d_in + 3 * BN + (size_t)b * 3 * N, P->normals, sizeof(double) * 3 * N)) return rc;
    if (int rc = up(d_in + 6 * BN + (size_t)b * 3 * N, P->normals_var, sizeof(double) * 3 * N)) return rc;
    if (int rc = up(U.d_valid + (size_t)b * N, P->valid, N)) return rc;
  }
  if (int rc = up(D.spd, spd.data(), sizeof(double) * BN)) return rc;
  if (int rc = up(D.spb, spb.data(), sizeof(double) * BN)) return rc;
  if (int rc = up(U.d_K, Kh.data(), sizeof(double) * 4 * B)) return rc;
  if (int rc = up(U.d_off, U.sp_off.data(), sizeof(int32_t) * ((size_t)B + 1))) return rc;
  INT_TRY(hipMemsetAsync(D.p0, 0, sizeof(double) * 2 * BN, st));
  if (!ids.empty()) {
    if (int rc = up(U.d_ids, ids.data(), sizeof(int32_t) * ids.size())) return rc;
    if (int rc = up(U.d_sp, U.sprec.data(), sizeof(double) * ids.size())) return rc;
    if (int rc = up(U.d_sp + ids.size(), U.sdep.data(), sizeof(double) * ids.size())) return rc;
  }
  U.act.assign((size_t)B, 1);
  if (int rc = up(U.d_act, U.act.data(), sizeof(int32_t) * B)) return rc;
  // everything staged must be on the device before the first kernel on st (and spd / spb / Kh die with this frame)
  if (int rc = staged_drain()) return rc;
  INT_TRY(hipStreamSynchronize(st));
  double* hblk = pool.get_host<double>((size_t)B * G * 8 + (size_t)B * 9);
  if (!hblk) return ifail(MPSFM_ENOMEM, "hipHostMalloc failed");
  U.hpart = hblk; U.hstate = hblk + (size_t)B * G * 8; U.hesp = U.hstate + (size_t)B * 8;
  return 0;
}

static int int_set_active(IntBatch& U, hipStream_t st) {
  INT_TRY(hipMemcpyAsync(U.d_act, U.act.data(), sizeof(int32_t) * U.B, hipMemcpyHostToDevice, st));
  INT_TRY(hipStreamSynchronize(st));  // U.act may change right after
  return 0;
}

static void int_launch_prepare(const mpsfm_int_problem* P0, IntBatch& U, hipStream_t st) {
  const size_t BN = (size_t)U.B * U.N;
  PrepArgs pa{P0->H, P0->W, U.d_in, U.d_in + BN, U.d_in + 3 * BN, U.d_in + 6 * BN, U.d_in + 2 * BN, U.d_valid, U.d_K,
              P0->large_number, P0->depth_magnitude_multiplier, P0->normals_magnitude_multiplier};
  hipLaunchKernelGGL(k_int_prepare, dim3(U.G, U.B), dim3(kIT), 0, st, pa, U.D);
}

// preconditioned CG with scipy.sparse.linalg.cg semantics (x0 = D.z, M = 1/clip(diag), rtol) on every
// active image; an image stops updating when its own test is met.  The host looks at the done flags every
// 16 iterations.  its[b] / conv[b] are written for the active images.
static int int_run_cg(IntBatch& U, hipStream_t st, double rtol, int max_iter, int* its, bool* conv) {
  IntDev& D = U.D;
  const int pix = ((size_t)U.B * U.N >= kPixSwitch) ? 2 : 1;
  const int Gc = (int)((U.N + (size_t)kIT * pix - 1) / ((size_t)kIT * pix));  // workgroups (= partial rows) of the CG kernels
  const dim3 grid(Gc, U.B);
  INT_TRY(hipMemsetAsync(D.state, 0, sizeof(double) * 8 * U.B, st));
  if (pix == 2) hipLaunchKernelGGL(k_cg_init<2>, grid, dim3(kIT), 0, st, D);
  else hipLaunchKernelGGL(k_cg_init<1>, grid, dim3(kIT), 0, st, D);
  int k = 0;
  bool done = false;
  while (!done && k < max_iter) {
    const int batch = std::min(16, max_iter - k);
    for (int j = 0; j < batch; ++j, ++k) {
      if (pix == 2) {
        hipLaunchKernelGGL(k_cg_dir<2>, grid, dim3(kIT), 0, st, D, Gc, k, k == 0 ? 1 : 0, rtol);
        hipLaunchKernelGGL(k_cg_update<2>, grid, dim3(kIT), 0, st, D, Gc, k);
      } else {
        hipLaunchKernelGGL(k_cg_dir<1>, grid, dim3(kIT), 0, st, D, Gc, k, k == 0 ? 1 : 0, rtol);
        hipLaunchKernelGGL(k_cg_update<1>, grid, dim3(kIT), 0, st, D, Gc, k);
      }
    }
    INT_TRY(hipMemcpyAsync(U.hstate, D.state, sizeof(double) * 8 * U.B, hipMemcpyDeviceToHost, st)); INT_TRY(hipStreamSynchronize(st));
    done = true;
    for (int b = 0; b < U.B; ++b)
      if (U.act[(size_t)b]) {
        const bool db = U.hstate[(size_t)b * 8 + 2] != 0.0;
        done = done && db;
        its[b] = (int)U.hstate[(size_t)b * 8 + 3];
        if (conv) conv[b] = db;
      }
  }
  INT_TRY(hipGetLastError());
  return 0;
}

}  // namespace mpsfm

using namespace mpsfm;

extern "C" int mpsfm_integrate_depth_batch(int32_t n_images, const mpsfm_int_problem* Ps, int32_t device, double* const* depth_out,
                                           mpsfm_int_summary* Ss) {
  if (n_images < 0) return ifail(MPSFM_EINVAL, "negative batch size");
  if (n_images == 0) return 0;
  if (!Ps || !depth_out || !Ss) return ifail(MPSFM_EINVAL, "NULL argument");
  const int B = n_images;
  const mpsfm_int_problem* P = &Ps[0];  // the shared configuration
  if (P->max_iter < 0 || P->max_iter > MPSFM_INT_MAX_IRLS) return ifail(MPSFM_EINVAL, "max_iter out of range");
  for (int b = 0; b < B; ++b) {
    if (!depth_out[b]) return ifail(MPSFM_EINVAL, "depth_out entry is NULL");
    if (int rc = int_check(&Ps[b], device)) return rc;
    if (!int_same_config(Ps[0], Ps[b])) return ifail(MPSFM_EINVAL, "images of a batch must share the map size and the configuration");
  }
  INT_TRY(hipSetDevice(device));
  std::memset(Ss, 0, sizeof(*Ss) * (size_t)B);
  // a stream of its own: concurrent calls from different host threads overlap on the GPU
  // U before the guard: destructors run in reverse order, so every early `return rc` first waits for the stream
  // (the guard) and only then hands U's device blocks back to the caching allocator
  IntBatch U;
  StreamGuard sg;
  INT_TRY(pooled_stream(&sg.st));
  hipStream_t st = sg.st;
  const size_t N = (size_t)P->H * P->W;
  if (int rc = int_setup(Ps, B, true, P->scale_filter != 0, st, U)) return rc;
  IntDev& D = U.D;
  const dim3 grid(U.G, B);
  std::vector<int32_t> keep((size_t)B, 0);
  for (int b = 0; b < B; ++b) {
    keep[(size_t)b] = (Ps[b].init && Ps[b].integrated && Ps[b].wu && Ps[b].wv) ? 1 : 0;
    if (keep[(size_t)b]) {
      INT_TRY(hipMemcpyAsync(D.wu + (size_t)b * N, Ps[b].wu, sizeof(double) * N, hipMemcpyHostToDevice, st));
      INT_TRY(hipMemcpyAsync(D.wv + (size_t)b * N, Ps[b].wv, sizeof(double) * N, hipMemcpyHostToDevice, st));
    }
  }
  INT_TRY(hipMemcpyAsync(U.d_keep, keep.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
  INT_TRY(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  INT_TRY(hipEventCreate(&e0)); INT_TRY(hipEventCreate(&e1));
  INT_TRY(hipEventRecord(e0, st));
  int_launch_prepare(P, U, st);

  // energies of the active images -> en[b]
  std::vector<double> en((size_t)B, 0.0);
  auto energy = [&]() -> int {
    hipLaunchKernelGGL(k_int_weights, grid, dim3(kIT), 0, st, D, P->k, P->lambda1, (const int32_t*)U.d_keep);
    hipLaunchKernelGGL(k_int_sparse_energy, dim3(B), dim3(kIT), 0, st, D, (const int32_t*)U.d_off, (const int32_t*)U.d_ids, (const double*)U.d_sp,
                       (const double*)(U.d_sp + U.ids.size()), P->lambda2, U.d_esp);
    INT_TRY(hipMemcpyAsync(U.hpart, D.part, sizeof(double) * (size_t)B * U.G * 8, hipMemcpyDeviceToHost, st));
    INT_TRY(hipMemcpyAsync(U.hesp, U.d_esp, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    INT_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      if (!U.act[(size_t)b]) continue;
      double e_n = 0.0, e_d = 0.0;
      const double* hp = U.hpart + (size_t)b * U.G * 8;
      for (int i = 0; i < U.G; ++i) { e_n += hp[(size_t)i * 8]; e_d += hp[(size_t)i * 8 + 1]; }
      const bool has_sparse = U.sp_off[(size_t)b + 1] > U.sp_off[(size_t)b];
      en[(size_t)b] = e_n + e_d + (has_sparse ? U.hesp[(size_t)b] : 0.0);
    }
    return 0;
  };
  auto finish = [&](int rc) {
    float ms = 0.f;
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    for (int b = 0; b < B; ++b) Ss[b].ms = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
  };
  auto save_weights = [&]() -> int {
    for (int b = 0; b < B; ++b) {
      if (Ps[b].wu) INT_TRY(hipMemcpyAsync(Ps[b].wu, D.wu + (size_t)b * N, sizeof(double) * N, hipMemcpyDeviceToHost, st));
      if (Ps[b].wv) INT_TRY(hipMemcpyAsync(Ps[b].wv, D.wv + (size_t)b * N, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    }
    INT_TRY(hipStreamSynchronize(st));
    return 0;
  };

  if (int rc = energy()) return finish(rc);
  // after the first evaluation the weights are always recomputed (update_W after every CG solve)
  std::fill(keep.begin(), keep.end(), 0);
  INT_TRY(hipMemcpyAsync(U.d_keep, keep.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
  std::vector<double> energy_0((size_t)B), min_energy((size_t)B);
  std::vector<uint8_t> success((size_t)B, 1), ran((size_t)B, 0);
  for (int b = 0; b < B; ++b) {
    mpsfm_int_summary& S = Ss[b];
    S.energy_initial = S.energy_final = en[(size_t)b];
    S.energies[0] = en[(size_t)b];
    S.energy_old_out = Ps[b].energy_old;
    S.integrated_out = Ps[b].integrated;
    energy_0[(size_t)b] = min_energy[(size_t)b] = en[(size_t)b];
    // energy has not changed: skip this frame (:433-437)
    if (Ps[b].integrated && !(std::fabs(en[(size_t)b] - Ps[b].energy_old) / Ps[b].energy_old > Ps[b].tol)) U.act[(size_t)b] = 0;
    else ran[(size_t)b] = 1;
  }
  std::vector<int> cg_its((size_t)B, 0);
  for (int it = 0; it < P->max_iter; ++it) {
    bool any = false;
    for (int b = 0; b < B; ++b) any = any || U.act[(size_t)b];
    if (!any) break;
    if (int rc = int_set_active(U, st)) return finish(rc);
    hipLaunchKernelGGL(k_int_system, grid, dim3(kIT), 0, st, D, P->lambda1);
    if (int rc = int_run_cg(U, st, P->cg_tol, P->cg_max_iter, cg_its.data(), nullptr)) return finish(rc);
    std::vector<double> energy_old(en);
    for (int b = 0; b < B; ++b) if (U.act[(size_t)b]) min_energy[(size_t)b] = std::min(en[(size_t)b], min_energy[(size_t)b]);
    if (int rc = energy()) return finish(rc);
    for (int b = 0; b < B; ++b) {
      if (!U.act[(size_t)b]) continue;
      mpsfm_int_summary& S = Ss[b];
      S.cg_iters[it] = cg_its[(size_t)b];
      S.cg_iterations_total += cg_its[(size_t)b];
      S.irls_iterations = it + 1;
      S.energies[it + 1] = en[(size_t)b];
      const double e = en[(size_t)b], eo = energy_old[(size_t)b], em = min_energy[(size_t)b];
      const double rel = std::fabs(e - eo) / eo, rel_min = std::fabs(e - em) / em;
      if (((rel < P->tol && (eo - e) > 0) || (rel_min < P->tol && (em - e) > 0)) && e < energy_0[(size_t)b]) U.act[(size_t)b] = 0;
      else if (e > energy_0[(size_t)b]) { success[(size_t)b] = 0; U.act[(size_t)b] = 0; }
    }
  }
  if (int rc = save_weights()) return finish(rc);
  bool any_changed = false;
  for (int b = 0; b < B; ++b) {
    mpsfm_int_summary& S = Ss[b];
    if (!ran[(size_t)b]) continue;  // skipped frame: summary already says so
    S.energy_final = en[(size_t)b];
    S.integrated_out = 1;
    if (!success[(size_t)b]) { S.energy_old_out = energy_0[(size_t)b]; continue; }  // energy increased: keep the old map (:504-508)
    S.energy_old_out = en[(size_t)b];
    S.changed = 1;
    any_changed = true;
  }
  if (any_changed) {
    const size_t BN = (size_t)B * N;
    hipLaunchKernelGGL(k_int_exp, dim3((unsigned)((BN + kIT - 1) / kIT)), dim3(kIT), 0, st, BN, (const double*)D.z, U.d_out);
    for (int b = 0; b < B; ++b)
      if (Ss[b].changed) INT_TRY(hipMemcpyAsync(depth_out[b], U.d_out + (size_t)b * N, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    INT_TRY(hipStreamSynchronize(st));
  }
  return finish(0);
}

extern "C" int mpsfm_integrate_depth(const mpsfm_int_problem* P, int32_t device, double* depth_out, mpsfm_int_summary* S) {
  if (!P || !depth_out || !S) return ifail(MPSFM_EINVAL, "NULL argument");
  double* outs[1] = {depth_out};
  return mpsfm_integrate_depth_batch(1, P, device, outs, S);
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
  IntBatch U;  // before the guard, see mpsfm_integrate_depth_batch
  StreamGuard sg;
  INT_TRY(pooled_stream(&sg.st));
  hipStream_t st = sg.st;
  const int N = P->H * P->W;
  if (int rc = int_setup(P, 1, use_sparse != 0, false, st, U)) return rc;  // calculate_hessian applies no scale filter (:542)
  IntDev& D = U.D;
  const dim3 grid(U.G, 1);
  int32_t zero = 0;
  INT_TRY(hipMemcpyAsync(U.d_keep, &zero, sizeof(int32_t), hipMemcpyHostToDevice, st));
  INT_TRY(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  INT_TRY(hipEventCreate(&e0)); INT_TRY(hipEventCreate(&e1));
  INT_TRY(hipEventRecord(e0, st));
  int_launch_prepare(P, U, st);
  hipLaunchKernelGGL(k_int_weights, grid, dim3(kIT), 0, st, D, P->k, P->lambda1, (const int32_t*)U.d_keep);  // init=False: weights from the checkpoint
  hipLaunchKernelGGL(k_int_system, grid, dim3(kIT), 0, st, D, P->lambda1);
  hipLaunchKernelGGL(k_int_unit_rhs, grid, dim3(kIT), 0, st, D);
  int its = 0;
  bool conv = false;
  int rc = int_run_cg(U, st, rtol, max_iter, &its, &conv);
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
