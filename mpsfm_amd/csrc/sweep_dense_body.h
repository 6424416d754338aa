// Body of the dense track sweep (see sweep_dense.hip for the phases), shared by k_track_sweep_dense and the single-launch
// solver of small problems (local_lm.hip).
#pragma once
#include "common.h"
#include "sweep_common.h"

namespace mpsfm {

namespace {
constexpr int kAccU = 2 * kDenseCams * 21;  // 2 copies x 16 cameras or 4 copies x 8 cameras
constexpr int kAccG = 2 * kDenseCams * 6;

}  // namespace

// Schur products of a chunk with at most 7 cameras (three row tiles; 4 x 28 blocks of partial sums fit the Z rows' LDS): the
// UNITS (landmark group, coordinate) are dealt to the four waves, every wave sums all tile pairs over its units — each Z row is
// fetched by exactly one wave, a quarter of the operand traffic of the pair-per-wave form, and the waves finish together — then
// the four partial sums meet in LDS (plain stores into a copy per wave over the no longer needed Z rows) and all threads write
// U - sum to the slab along its 288-byte blocks.
template <int NT>
__device__ __forceinline__ void schur_units(double* s_W, const uint8_t* s_rec, const double* s_U, double* slab, int tid, int npt, int ncam,
                                            int ncopy, int cstride, int dbg) {
  constexpr int NP = NT * (NT + 1) / 2;
  const int lane = tid & 63, wave = tid >> 6;
  const int rc = lane & 15, kq = lane >> 4;
  int camt[NT], a3t[NT];
  bool camv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = 16 * t + rc;
    camt[t] = r / 6; a3t[t] = (r - 6 * (r / 6)) * 3;
    camv[t] = camt[t] < ncam; camt[t] = camv[t] ? camt[t] : 0;
  }
  v4d acc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
  const int nunits = 3 * ((npt + 3) >> 2);
  if (!(dbg & 2)) {
    // two-stage pipeline over the wave's units u = wave, wave + 4, ...: record indices one unit ahead, then the operand
    int rcur[NT], rnxt[NT];
    double x[NT];
    auto rec_request = [&](int u, int (&r)[NT]) {
      const int pp = min(4 * (u / 3) + kq, npt - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) r[t] = (int)s_rec[pp * kDenseCams + camt[t]];
    };
    auto rec_fix = [&](int u, int (&r)[NT]) {
      const bool pv = 4 * (u / 3) + kq < npt;
#pragma unroll
      for (int t = 0; t < NT; ++t) r[t] = (pv && camv[t]) ? r[t] : 255;
    };
    int u = wave;
    if (u < nunits) {
      rec_request(u, rcur);
      rec_fix(u, rcur);
#pragma unroll
      for (int t = 0; t < NT; ++t) x[t] = s_W[(rcur[t] == 255 ? 0 : rcur[t]) * kWStride + a3t[t] + u % 3];
      rec_request(min(u + 4, nunits - 1), rnxt);
#pragma unroll
      for (int t = 0; t < NT; ++t) x[t] = rcur[t] == 255 ? 0.0 : x[t];
    }
    for (; u < nunits; u += 4) {
      const int u1 = min(u + 4, nunits - 1), u2 = min(u + 8, nunits - 1);
      double xn[NT];
      rec_fix(u1, rnxt);
#pragma unroll
      for (int t = 0; t < NT; ++t) xn[t] = s_W[(rnxt[t] == 255 ? 0 : rnxt[t]) * kWStride + a3t[t] + u1 % 3];
#pragma unroll
      for (int t = 0; t < NT; ++t) rcur[t] = rnxt[t];
      rec_request(u2, rnxt);
      {
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int tj = ti; tj < NT; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ti], x[tj], acc[q], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) x[t] = rcur[t] == 255 ? 0.0 : xn[t];
    }
  }
  const int nb = ncam * (ncam + 1) / 2;
  __syncthreads();  // every wave is done with the Z rows
  {
    double* part = s_W + wave * (nb * 36);
    int q = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = ti; tj < NT; ++tj, ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + kq + 4 * r, col = 16 * tj + rc;
          const int ci = row / 6, cj = col / 6;
          if (cj < ncam && ci <= cj) part[(cj * (cj + 1) / 2 + ci) * 36 + (row - 6 * ci) * 6 + (col - 6 * cj)] = acc[q][r];
        }
  }
  __syncthreads();
  if (!(dbg & 4))
    for (int idx = tid; idx < nb * 36; idx += kThreads) {
      const int b = idx / 36, el = idx - b * 36;
      int cj = (int)((sqrtf(8.0f * (float)b + 1.0f) - 1.0f) * 0.5f);
      while (cj * (cj + 1) / 2 > b) --cj;
      while ((cj + 1) * (cj + 2) / 2 <= b) ++cj;
      const int ci = b - cj * (cj + 1) / 2;
      const int ra = el / 6, cb = el - ra * 6;
      double v = -((s_W[idx] + s_W[nb * 36 + idx]) + (s_W[2 * nb * 36 + idx] + s_W[3 * nb * 36 + idx]));
      if (ci == cj && cb >= ra) {  // the camera's own U block rides on its diagonal Schur block (upper triangle; the lower one is not read)
        const int uix = ra * 6 - (ra * (ra - 1)) / 2 + (cb - ra);
        for (int c = 0; c < ncopy; ++c) v += s_U[(c * cstride + ci) * 21 + uix];
      }
      slab[idx] = v;
    }
}

// The wave's share of the chunk's Schur products.  Pairs (ti <= tj) of 16-row tiles in row-major order of the upper triangle;
// the wave with (wave + chunk) % 4 == w takes q = w, w + 4, ...: the waves with one pair more change from chunk to chunk, so that
// the four matrix pipes of a CU see the same load.  NP: the pairs this call sums (accumulators are register arrays: static),
// starting with the wave's pair number `first`.
//
// The loop over landmark groups is a three-stage pipeline written out by hand, without a branch inside: at the top of iteration g
// the Z rows of group g + 1 are requested (their record indices arrived during iteration g - 1), then the record indices of
// group g + 2 (s_rec), then the products of group g are issued — so no LDS round trip is exposed; the compiler's own schedule
// waited for every s_rec byte before requesting the row behind it, four dependent round trips per iteration.  Two
// accumulators per pair (coordinates 0, 2 and 1) keep dependent matrix instructions apart.
template <int NP>
__device__ __forceinline__ void schur_pairs(const double* s_W, const uint8_t* s_rec, const double* s_U, double* slab, int tid, int cix, int npt, int ncam,
                                            int NT, int ncopy, int cstride, int dbg, int first) {
  const int lane = tid & 63, wave = tid >> 6;
  const int rc = lane & 15, kq = lane >> 4;
  const int wq = (wave + cix) & 3;
  constexpr int NO = 2 * NP;  // operand slots: A and B of every pair (B repeats A on a diagonal pair)
  int pti[NP], ptj[NP], camo[NO], a3o[NO];
  bool camv[NO];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int ti = 0, rem = wq + 4 * (first + j);  // wave-uniform decode
    while (ti < NT - 1 && rem >= NT - ti) { rem -= NT - ti; ++ti; }
    const int tj = min(ti + rem, NT - 1);
    pti[j] = ti; ptj[j] = tj;
    const int ra = 16 * ti + rc, rb = 16 * tj + rc;
    camo[2 * j] = ra / 6; a3o[2 * j] = (ra - 6 * (ra / 6)) * 3;
    camo[2 * j + 1] = rb / 6; a3o[2 * j + 1] = (rb - 6 * (rb / 6)) * 3;
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) { camv[o] = camo[o] < ncam; camo[o] = camv[o] ? camo[o] : 0; }
  v4d acc[NP][2];
#pragma unroll
  for (int j = 0; j < NP; ++j) { acc[j][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[j][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
  const int ng = (npt + 3) >> 2;
  if (!(dbg & 2)) {
    int rcur[NO], rnxt[NO];
    double x[NO][3], xn[NO][3];
    auto rec_request = [&](int g, int (&r)[NO]) {  // clamped addresses: always a valid cell, masked in rec_fix
      const int pp = min(4 * g + kq, npt - 1);
#pragma unroll
      for (int o = 0; o < NO; ++o) r[o] = (int)s_rec[pp * kDenseCams + camo[o]];
    };
    auto rec_fix = [&](int g, int (&r)[NO]) {
      const bool pv = 4 * g + kq < npt;
#pragma unroll
      for (int o = 0; o < NO; ++o) r[o] = (pv && camv[o]) ? r[o] : 255;
    };
    auto row_request = [&](const int (&r)[NO], double (&v)[NO][3]) {
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const double* z = &s_W[(r[o] == 255 ? 0 : r[o]) * kWStride + a3o[o]];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[o][c] = z[c];
      }
    };
    auto row_fix = [&](const int (&r)[NO], double (&v)[NO][3]) {
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[o][c] = r[o] == 255 ? 0.0 : v[o][c];
    };
    rec_request(0, rcur);
    rec_fix(0, rcur);
    row_request(rcur, x);
    rec_request(1 < ng ? 1 : 0, rnxt);
    row_fix(rcur, x);
    for (int g = 0; g < ng; ++g) {
      const int g1 = min(g + 1, ng - 1), g2 = min(g + 2, ng - 1);  // past the end: repeats of the last group, never used
      rec_fix(g1, rnxt);
      row_request(rnxt, xn);
#pragma unroll
      for (int o = 0; o < NO; ++o) rcur[o] = rnxt[o];
      rec_request(g2, rnxt);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < NP; ++j)
          acc[j][c & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2 * j][c], x[2 * j + 1][c], acc[j][c & 1], 0, 0, 0);
      row_fix(rcur, xn);
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int c = 0; c < 3; ++c) x[o][c] = xn[o][c];
    }
  }
  // accumulator element r of lane (kq, rc): row kq + 4 r, column rc of its tile pair
  if (!(dbg & 4)) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * pti[j] + kq + 4 * r, col = 16 * ptj[j] + rc;
        const int ci = row / 6, cj = col / 6;
        const int ra = row - 6 * ci, cb = col - 6 * cj;
        if (cj < ncam && (ci < cj || (ci == cj && ra <= cb))) {
          double v = -(acc[j][0][r] + acc[j][1][r]);
          if (ci == cj) {  // the camera's own U block rides on its diagonal Schur block (upper triangle)
            const int u = ra * 6 - (ra * (ra - 1)) / 2 + (cb - ra);
            double uu = 0.0;
            for (int q = 0; q < ncopy; ++q) uu += s_U[(q * cstride + ci) * 21 + u];
            v += uu;
          }
          slab[(cj * (cj + 1) / 2 + ci) * 36 + ra * 6 + cb] = v;
        }
      }
    }
  }
}


// LDS of one dense chunk's sweep: 53.3 KB
struct __attribute__((aligned(16))) DenseLds {
  double W[kObsMax * kWStride];
  double V[kDensePts * 6];
  double g[kDensePts * 3];
  double U[kAccU];
  double gc[kAccG];
  double wv[kAccG];
  uint8_t rec[kDensePts * kDenseCams] __attribute__((aligned(4)));  // record of (landmark, local camera), 255: none
  int32_t slot[kDenseCams];
  double red[3 * (kThreads / 64)];
};
static_assert(kObsMax <= 254, "record indices in DenseLds::rec are 8 bits, 255 marks an empty cell");
static_assert(sizeof(DenseLds) <= 54608, "three workgroups per CU need at most 160 KiB / 3 of LDS each");

// The sweep of chunk `cix` by one workgroup of kThreads threads: slab and partial row written, see sweep_dense.hip.
template <bool kLocal>
__device__ __forceinline__ void dense_sweep_chunk(const SweepArgs& A, int cix, double lm_radius, const double* l_tab, DenseLds& S) {
  const int tid = thread_index<kLocal>();
  // timing trace (dbg flag 128, scripts/dbg_sweep_trace.py): lane 0 of every wave stamps the shader clock at the phase boundaries
  // into diagV, which only the Jacobi-scaling pass uses (16 slots x 4 waves per chunk)
  long long* const trace = (A.dbg & 128) && (tid & 63) == 0 ? reinterpret_cast<long long*>(A.diagV) + ((size_t)cix * 4 + (tid >> 6)) * 16 : nullptr;
#define MPSFM_STAMP(k) do { if (trace) trace[k] = (long long)clock64(); } while (0)
  MPSFM_STAMP(0);
  if (trace) trace[12] = (long long)wall_clock64();  // 100 MHz reference, to calibrate the shader clock
  const ChunkHdr H = A.chunks[cix];
  const int nrec = H.nrec, npt = H.npt, ncam = H.ncam;
  const int ncopy = ncam <= 8 ? 4 : 2, cstride = ncam <= 8 ? 8 : 16;  // accumulator copies x cameras per copy = 32

  // ---- P0 ------------------------------------------------------------------------------------------------------------
  // first launch of an iteration: the candidate the last decision accepted is what this sweep linearises at; its copy into the
  // state (this chunk's landmarks; the cameras and the zeroing of the reduced buffer dealt over all workgroups) runs beside P1
  bool adopt = false;
  if constexpr (!kLocal) {
    if (A.adopt_on) {
      adopt = lm_accepted(A.ctl);
      const int64_t g0 = (int64_t)(cix - A.chunk0) * kThreads + tid, gstep = (int64_t)A.nchunks * kThreads;
      for (int64_t e = g0; e < A.nred; e += gstep) A.red[e] = 0.0;
      if (adopt) {
        for (int i = tid; i < 3 * npt; i += kThreads) A.pts_rw[(size_t)3 * H.pt0 + i] = A.pts2[(size_t)3 * H.pt0 + i];
        const int64_t n_q = 4 * (int64_t)A.adopt_nc, n_t = 3 * (int64_t)A.adopt_nc, n_tab = (int64_t)kCamRec * A.adopt_nc;
        for (int64_t e = g0; e < n_q + n_t + n_tab; e += gstep) {
          if (e < n_q) A.q_rw[e] = A.q2[e];
          else if (e < n_q + n_t) A.t_rw[e - n_q] = A.t2[e - n_q];
          else A.camtab_rw[e - n_q - n_t] = A.camtab2[e - n_q - n_t];
        }
      }
    }
  }
  const double* const pts_at = adopt ? A.pts2 : A.pts;
  const double* const camtab_at = adopt ? A.camtab2 : A.camtab;
  for (int i = tid; i < npt * 6; i += kThreads) S.V[i] = 0.0;
  for (int i = tid; i < npt * 3; i += kThreads) S.g[i] = 0.0;
  for (int i = tid; i < kAccU; i += kThreads) S.U[i] = 0.0;
  for (int i = tid; i < kAccG; i += kThreads) { S.gc[i] = 0.0; S.wv[i] = 0.0; }
  for (int i = tid; i < npt * (kDenseCams / 4); i += kThreads) reinterpret_cast<uint32_t*>(S.rec)[i] = 0xffffffffu;
  if (tid < ncam) S.slot[tid] = A.chunk_cams[H.cam0 + tid];
  MPSFM_STAMP(1);
  __syncthreads();
  MPSFM_STAMP(2);

  // ---- P1 ------------------------------------------------------------------------------------------------------------
  double my_cost = 0.0;
  int my_bad = 0;
  uint32_t my_meta = 0;
  double my_w[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) my_w[i] = 0.0;
  double Vg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // this record's share of V_p (packed upper triangle) and g_p
  if (tid < nrec) {
    const int rix = H.rec0 + tid;
    const uint32_t meta = A.rec_meta[rix];
    my_meta = meta;
    const int cam = A.rec_cam[rix];
    const int lcam = meta & 0xff;
    const int lpt = (meta >> 8) & 0xff;
    const double2 xy = reinterpret_cast<const double2*>(A.rec_xy)[rix];
    double d = 1.0, m = 0.0, a = 1.0;
    if (meta & kRecHasDepth) { d = A.rec_d[rix]; m = A.rec_m[rix]; a = A.rec_a[rix]; }
    const int pix = H.pt0 + lpt;
    const double X[3] = {pts_at[3 * pix], pts_at[3 * pix + 1], pts_at[3 * pix + 2]};
    const double psc[3] = {A.ps[3 * pix], A.ps[3 * pix + 1], A.ps[3 * pix + 2]};
    RecLin L;
    if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace[10] = (long long)clock64(); }  // the record / landmark loads have landed (the camera row follows)
    linearize_record(camera_row<kLocal>(camtab_at, l_tab, S.slot, cam, lcam), X, psc, meta, xy.x, xy.y, d, m, a, A.loss, L);
    my_cost = L.cost;
    if (trace) { asm volatile("" : "+v"(L.Jc[0]), "+v"(L.Jp[8]), "+v"(L.cost)); trace[11] = (long long)clock64(); }
    my_bad = L.ok ? 0 : 1;
    if (L.ok) {
      if (psc[0] != 0.0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double j0 = L.Jp[3 * r], j1 = L.Jp[3 * r + 1], j2 = L.Jp[3 * r + 2];
          Vg[0] += j0 * j0; Vg[1] += j0 * j1; Vg[2] += j0 * j2; Vg[3] += j1 * j1; Vg[4] += j1 * j2; Vg[5] += j2 * j2;
          Vg[6] += j0 * L.r[r]; Vg[7] += j1 * L.r[r]; Vg[8] += j2 * L.r[r];
        }
      }
      if (lcam != (int)kLcamConst && !(A.dbg & 1)) {
        const int acc = (lpt % ncopy) * cstride + lcam;
        int u = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double gci = L.Jc[i] * L.r[0] + L.Jc[6 + i] * L.r[1] + L.Jc[12 + i] * L.r[2];
          atomicAdd(&S.gc[acc * 6 + i], gci);
#pragma unroll
          for (int j = i; j < 6; ++j, ++u) {
            const double uij = L.Jc[i] * L.Jc[j] + L.Jc[6 + i] * L.Jc[6 + j] + L.Jc[12 + i] * L.Jc[12 + j];
            atomicAdd(&S.U[acc * 21 + u], uij);
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            my_w[i * 3 + j] = L.Jc[i] * L.Jp[j] + L.Jc[6 + i] * L.Jp[3 + j] + L.Jc[12 + i] * L.Jp[6 + j];
      }
    }
  }
  // Landmark side: the records of a landmark are NEIGHBOURING lanes, so their nine sums would meet on one LDS address (~5 lanes per
  // address and instruction).  They are first summed along the lanes — a segmented scan over lanes with equal landmark, by DPP row
  // shifts inside the 16-lane rows (vector pipe, no LDS) — and only the last lane of every (row, landmark) run adds to LDS.  All
  // lanes take part: threads beyond the records carry zeros and landmark 0.
  {
    const int my_lpt = (int)((my_meta >> 8) & 0xff);
    seg_step<1>(my_lpt, Vg); seg_step<2>(my_lpt, Vg); seg_step<4>(my_lpt, Vg); seg_step<8>(my_lpt, Vg);
    const int nlpt = __builtin_amdgcn_update_dpp(-1, my_lpt, 0x101, 0xf, 0xf, false);  // row_shl:1: the right neighbour's landmark
    if (nlpt != my_lpt && my_lpt < npt) {  // last lane of its run inside the row (lane 15 of a row sees -1)
#pragma unroll
      for (int k = 0; k < 6; ++k) if (Vg[k] != 0.0) atomicAdd(&S.V[my_lpt * 6 + k], Vg[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) if (Vg[6 + k] != 0.0) atomicAdd(&S.g[my_lpt * 3 + k], Vg[6 + k]);
    }
  }
  MPSFM_STAMP(3);
  __syncthreads();
  MPSFM_STAMP(4);

  // ---- P2 + P3a: every record factors ITS landmark's 3x3 block itself — F = chol(V + D)^-1, the ~5 records of a landmark repeat
  // the same ~60 operations — instead of one thread per landmark and another barrier; then Z = W F^T to LDS and W V^-1 g_p per
  // camera.  The landmark's first record reports its gradient maximum and a failed factorisation.
  double my_gmax = 0.0;
  const int prev_lpt = (__shfl_up((int)my_meta, 1, 64) >> 8) & 0xff;  // all lanes take part (threads beyond the records carry meta 0)
  if (tid < nrec) {
    const int lcam = my_meta & 0xff;
    const int lpt = (my_meta >> 8) & 0xff;
    const uint16_t kv = A.pt_kv[H.pt0 + lpt];
    if (kv != 0xffff) {  // variable landmark
      double V[6], F[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) V[k] = S.V[lpt * 6 + k];
      V[0] += fmin(fmax(V[0], A.min_diag), A.max_diag) / lm_radius;
      V[3] += fmin(fmax(V[3], A.min_diag), A.max_diag) / lm_radius;
      V[5] += fmin(fmax(V[5], A.min_diag), A.max_diag) / lm_radius;
      const bool okf = spd3_inv_factor(V, F);
      if (!okf) {
#pragma unroll
        for (int k = 0; k < 6; ++k) F[k] = 0.0;
      }
      const double g0 = S.g[lpt * 3], g1 = S.g[lpt * 3 + 1], g2 = S.g[lpt * 3 + 2];
      const bool first = (tid & 63) == 0 || prev_lpt != lpt;
      // (a landmark whose records straddle two waves reports twice: the maximum and the invalid flag do not mind)
      if (first) {
        if (!okf) my_bad = 1;
        const double p0 = A.ps[3 * (H.pt0 + lpt)], p1 = A.ps[3 * (H.pt0 + lpt) + 1], p2 = A.ps[3 * (H.pt0 + lpt) + 2];
        my_gmax = fmax(fabs(g0 / p0), fmax(fabs(g1 / p1), fabs(g2 / p2)));
      }
      if (lcam != (int)kLcamConst) {
        double* w = &S.W[tid * kWStride];
        const double v0 = F[0] * g0, v1 = F[1] * g0 + F[2] * g1, v2 = F[3] * g0 + F[4] * g1 + F[5] * g2;  // F g
        const int acc = (lpt % ncopy) * cstride + lcam;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double w0 = my_w[i * 3], w1 = my_w[i * 3 + 1], w2 = my_w[i * 3 + 2];
          const double z0 = w0 * F[0], z1 = w0 * F[1] + w1 * F[2], z2 = w0 * F[3] + w1 * F[4] + w2 * F[5];
          w[i * 3] = z0; w[i * 3 + 1] = z1; w[i * 3 + 2] = z2;
          atomicAdd(&S.wv[acc * 6 + i], z0 * v0 + z1 * v1 + z2 * v2);
        }
        if (my_bad == 0 && okf) S.rec[lpt * kDenseCams + lcam] = (uint8_t)tid;  // this record's Z takes part in the products
      }
    }
  }
  MPSFM_STAMP(5);
  __syncthreads();  // S.W holds Z, S.rec says where
  MPSFM_STAMP(6);

  // ---- P3b: the wave's tile pairs of M M^T over all landmarks, straight to the slab -----------------------------------------
  double* const slab = A.slab + (size_t)H.slab0 * 18;
  const int nb = ncam * (ncam + 1) / 2;
  {
    const int NT = (6 * ncam + 15) >> 4;
    static_assert(4 * 28 * 36 <= kObsMax * kWStride, "four copies of the blocks of 7 cameras must fit the Z rows");
    if (ncam <= 7 && ncam > 0 && !(A.dbg & 32)) {  // (workgroup-uniform: the barriers inside are safe)
      if (NT == 1) schur_units<1>(S.W, S.rec, S.U, slab, tid, npt, ncam, ncopy, cstride, A.dbg);
      else if (NT == 2) schur_units<2>(S.W, S.rec, S.U, slab, tid, npt, ncam, ncopy, cstride, A.dbg);
      else schur_units<3>(S.W, S.rec, S.U, slab, tid, npt, ncam, ncopy, cstride, A.dbg);
    } else {
      const int npairs = NT * (NT + 1) / 2;
      const int wq = ((tid >> 6) + cix) & 3;
      const int mine = npairs > wq ? (npairs - wq + 3) >> 2 : 0;  // this wave's pairs: at most 2 up to three row tiles (8 cameras), 6 beyond
      int done = 0;
      if constexpr (kLocal) {  // one workgroup per CU there: registers for four pairs per pass (nine and more cameras: one pass instead of two)
        for (; done + 4 <= mine; done += 4) schur_pairs<4>(S.W, S.rec, S.U, slab, tid, cix, npt, ncam, NT, ncopy, cstride, A.dbg, done);
      }
      for (; done + 2 <= mine; done += 2) schur_pairs<2>(S.W, S.rec, S.U, slab, tid, cix, npt, ncam, NT, ncopy, cstride, A.dbg, done);
      if (done < mine) schur_pairs<1>(S.W, S.rec, S.U, slab, tid, cix, npt, ncam, NT, ncopy, cstride, A.dbg, done);
    }
  }

  MPSFM_STAMP(7);
  // ---- P4: camera vectors to the slab, chunk partials ---------------------------------------------------------------------
  if (!(A.dbg & 4))
    for (int idx = tid; idx < ncam * 18; idx += kThreads) {  // 16 cameras x 18 > 256 threads
      const int lc = idx / 18, k = idx - lc * 18;
      double v = 0.0;
      if (k < 6) { for (int q = 0; q < ncopy; ++q) v += S.gc[(q * cstride + lc) * 6 + k]; }
      else if (k < 12) { for (int q = 0; q < ncopy; ++q) v += S.wv[(q * cstride + lc) * 6 + (k - 6)]; }
      else {
        const int i = k - 12, u = i * 6 - (i * (i - 1)) / 2;  // diagonal entry (i, i) of the packed upper triangle
        for (int q = 0; q < ncopy; ++q) v += S.U[(q * cstride + lc) * 21 + u];
      }
      slab[nb * 36 + idx] = v;
    }
  {
    const double c = wave_sum(my_cost);
    const double b = wave_sum((double)my_bad);
    const double g = wave_max(my_gmax);
    const int w = tid >> 6;
    if ((tid & 63) == 0) { S.red[w] = c; S.red[4 + w] = b; S.red[8 + w] = g; }
    MPSFM_STAMP(8);
    __syncthreads();
    MPSFM_STAMP(9);
    if (trace) trace[13] = (long long)wall_clock64();
    if (tid == 0) {
      double* p = A.part + (size_t)cix * 4;
      p[0] = (S.red[0] + S.red[1]) + (S.red[2] + S.red[3]);
      p[1] = (S.red[4] + S.red[5]) + (S.red[6] + S.red[7]);
      p[2] = fmax(fmax(S.red[8], S.red[9]), fmax(S.red[10], S.red[11]));
      p[3] = 0.0;
    }
  }
#undef MPSFM_STAMP
#undef MPSFM_STAMP
}

}  // namespace mpsfm
