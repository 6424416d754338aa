// Track sweep of the DENSE chunks (gfx950) and the reduction of their slabs.
//
// What Ceres does per LM iteration inside pyceres.solve for the problem of reference
// mpsfm/sfm/mapper/bundle_adjustment.py:67-185 — evaluate every residual block and its Jacobian, form J^T J, eliminate the
// landmark blocks (SPARSE_SCHUR) — for one chunk of consecutive (camera-set sorted) landmarks per workgroup.  A dense chunk
// has at most kDenseCams (16) variable cameras, kDensePts (96) landmarks, kObsMax (252) merged records and one record per
// (camera, variable landmark).
//
//   P0  clear the LDS accumulators, request the landmark data of P2
//   P1  one thread per record: residuals, analytic Jacobians, robust weights (sweep_common.h);
//       V_p, g_p per landmark and U_c (upper triangle), g_c per camera by ds_add_f64 — the camera accumulators in 4 (up to 8
//       cameras) or 2 copies chosen by the landmark, because neighbouring lanes are the records of one landmark and would
//       otherwise add ~13-fold to one address; W = Jc^T Jp stays in registers
//   P2  one thread per landmark: F = chol(V + D)^-1, F g_p
//   P3a Z = W F^T to LDS (18 doubles per record), W V^-1 g_p per camera
//   P3b S_chunk = M M^T, M the (6 ncam) x (3 npt) matrix of the Z blocks, on the matrix pipe: v_mfma_f64_16x16x4_f64 with a
//       LANDMARK as K index (four landmarks per instruction, one coordinate each).  The 16-row tile pairs (ti <= tj) are
//       dealt to the four waves; a wave sums ITS pairs over all landmarks, so every accumulator is complete in its registers
//       and goes straight to the chunk's SLAB in HBM with plain stores (U_c - Z Z^T on the diagonal blocks) — no LDS staging,
//       no LDS atomics, no global atomics
//   P4  g_c | W V^-1 g_p | diag U of every local camera to the slab, chunk partials (cost, invalid records, max |g_p|)
//
// k_reduce_slabs then sums the slabs per destination (a 6x6 block of S or a camera's vectors) and adds the sums into the
// reduced buffer: 0.7 MB of atomics at C3 instead of 24 MB, and a fixed summation order per destination part.
// LDS: 53.4 KB per workgroup = three workgroups per CU.
#include "sweep_dense_body.h"
#include <algorithm>

namespace mpsfm {

__global__ __launch_bounds__(kThreads, 3) void k_track_sweep_dense(SweepArgs A) {
  if (lm_over(A.ctl)) return;
  const double lm_radius = A.ctl ? lm_radius_of(A.ctl) : A.radius;
  __shared__ DenseLds S;
  // One workgroup per chunk.  Persistent workgroups (3 per CU taking chunks from a cost-sorted list through a shared counter) were
  // measured and dropped: the slots stay full, but every chunk then takes longer (wave life 21.5 -> 24.7 us at C3: the sweep is
  // bound by the CU's shared pipes, not by the 22 % of slot time the dispatcher leaves empty) — 0.118 against 0.111 ms.
  dense_sweep_chunk<false>(A, blockIdx.x + A.chunk0, lm_radius, nullptr, S);
}

// One wave per destination part: sums the part's slab sources in table order and adds the sum into the (zeroed) reduced buffer.
__global__ __launch_bounds__(256) void k_reduce_slabs(const RedDest* __restrict__ dests, int ndest, const int32_t* __restrict__ srcs,
                                                      const double* __restrict__ slab, double* Sblk, double* gc, double* wv, double* diagU,
                                                      const LmCtl* ctl) {
  if (lm_over(ctl)) return;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= ndest) return;
  const RedDest D = dests[w];
  bool active;
  if (D.kind == 2) active = lane < 18;
  else if (D.kind == 1) { const int ra = lane / 6, cb = lane - 6 * ra; active = lane < 36 && cb >= ra; }
  else active = lane < 36;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int s = D.s0;
  for (; s + 8 <= D.s1; s += 8) {
    int32_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = srcs[s + k];
    if (active) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += slab[(size_t)o[k] * 18 + lane];
    }
  }
  for (int k = 0; s < D.s1; ++s, ++k)
    if (active) acc[k & 7] += slab[(size_t)srcs[s] * 18 + lane];
  const double v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (!active || v == 0.0) return;
  if (D.kind == 2) {
    double* base = lane < 6 ? gc : (lane < 12 ? wv : diagU);
    atomicAdd(&base[(size_t)D.dst * 6 + (lane % 6)], v);
  } else {
    atomicAdd(&Sblk[(size_t)D.dst * 36 + lane], v);
    if (D.kind == 1) {  // nothing: diag U travels in the camera vectors
    }
  }
}

void launch_track_sweep_dense(const SweepArgs& a, int nchunks, hipStream_t s) {
  if (nchunks > 0) hipLaunchKernelGGL(k_track_sweep_dense, dim3(nchunks), dim3(kThreads), 0, s, a);
}
void launch_reduce_slabs(const RedDest* dests, int ndest, const int32_t* srcs, const double* slab, double* Sblk, double* gc, double* wv, double* diagU,
                         const LmCtl* ctl, hipStream_t s) {
  if (ndest > 0) hipLaunchKernelGGL(k_reduce_slabs, dim3((ndest + 3) / 4), dim3(256), 0, s, dests, ndest, srcs, slab, Sblk, gc, wv, diagU, ctl);
}

}  // namespace mpsfm
