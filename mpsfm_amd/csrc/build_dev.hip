// Device-side table build of the bundle-adjustment handle (gfx950): what `build()` in ba_solver.hip does on host threads —
// residual blocks grouped by landmark, merged into records, landmarks ordered by their camera-slot lists, the greedy chunk cut,
// the record arrays — as HIP kernels over the caller's raw observation lists.  The reference assembles a NEW problem for every
// Optimizer.ba() call (mpsfm/sfm/mapper/bundle_adjustment.py:184, 285-293), so the tables are on the critical path of every call.
//
// Two stages around the camera order (the graph / nested dissection / factorisation plan stay on the host, 0.3-0.5 ms):
//   stage 1  upload the observation lists; count blocks per landmark and per camera; scan; scatter the blocks to their
//            landmark; camera graph (bit matrix, test-before-atomicOr)                              -> host: graph, counts
//   stage 2  per landmark: sort its blocks by (slot, camera, kind, source), count the merged records, flags (long / heavy),
//            sort key; landmark order by two stable radix sorts (rocPRIM) + a stable class sort; record offsets by scan; the
//            chunk cut — one WAVE per segment, the open chunk's camera set a 4096-bit set spread over the 64 lanes —; chunk
//            headers and camera lists compacted; the record arrays written in place                  -> host: chunks, camera lists,
//            landmark order, counters
// The results are the tables of the host build BIT FOR BIT (tests/test_gpu_devbuild.py compares every table of two handles),
// except `rec_d` = log(depth), where the device's log may differ from libm's in the last place.
//
// Not covered (the caller falls back to the host threads): landmark-sharded runs (the exchanges of the build are host values),
// more than 4096 variable cameras (no camera graph), a landmark with more blocks than a chunk holds (long tracks).
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "common.h"
#include "devbuild.h"

namespace mpsfm {
extern thread_local std::string g_err;
int staged_upload(void* dst, const void* src, size_t bytes);
int staged_drain();
static int dfail(int code, const std::string& m) { g_err = m; return code; }
#define DB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return dfail(MPSFM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

namespace {
constexpr int kT = 256;
constexpr int kLmSlots = 32;  // distinct variable slots cached per landmark for the chunk cut (longer lists are re-read from the blocks)

struct LmInfo {  // per landmark (caller's index), written by k_sortmerge
  int32_t nrec, nfix, kv, distinct;
  uint32_t flags;  // 1 long, 2 heavy, 4 dup, 8 bad depth
  uint32_t pad;
  uint64_t k1, k2;
};
enum { LM_LONG = 1, LM_HEAVY = 2, LM_DUP = 4, LM_BAD = 8 };

__global__ __launch_bounds__(kT) void k_count(int64_t n_obs, int64_t n_dobs, const int32_t* obs_cam, const int32_t* obs_pt, const int32_t* dobs_cam,
                                              const int32_t* dobs_pt, const double* dobs_depth, int nc, int np, int32_t* cnt_pt, int32_t* cnt_cam, int32_t* err) {
  extern __shared__ int32_t s_cam[];
  for (int i = threadIdx.x; i < nc; i += kT) s_cam[i] = 0;
  __syncthreads();
  const int64_t n = n_obs + n_dobs;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
    int cam, pt;
    if (i < n_obs) { cam = obs_cam[i]; pt = obs_pt[i]; }
    else {
      cam = dobs_cam[i - n_obs]; pt = dobs_pt[i - n_obs];
      if (!(dobs_depth[i - n_obs] > 0.0)) atomicOr(err, 1);
    }
    atomicAdd(&cnt_pt[pt], 1);
    atomicAdd(&s_cam[cam], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += kT) if (s_cam[i]) atomicAdd(&cnt_cam[i], s_cam[i]);
}

__global__ __launch_bounds__(kT) void k_scatter(int64_t n_obs, int64_t n_dobs, const int32_t* obs_cam, const int32_t* obs_pt, const int32_t* dobs_cam,
                                                const int32_t* dobs_pt, const int32_t* pstart, int32_t* fill, int32_t* blk_cam, uint32_t* blk_src) {
  const int64_t n = n_obs + n_dobs;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
    int cam, pt;
    uint32_t src;
    if (i < n_obs) { cam = obs_cam[i]; pt = obs_pt[i]; src = (uint32_t)i; }
    else { cam = dobs_cam[i - n_obs]; pt = dobs_pt[i - n_obs]; src = (uint32_t)(i - n_obs) | 0x80000000u; }
    const int at = pstart[pt] + atomicAdd(&fill[pt], 1);
    blk_cam[at] = cam; blk_src[at] = src;
  }
}

// camera graph: two variable cameras are adjacent when a variable landmark has blocks of both (caller's "natural" slots)
__global__ __launch_bounds__(kT) void k_graph(int np, const int32_t* pstart, const int32_t* blk_cam, const uint8_t* pt_const, const int32_t* nat,
                                              int words, unsigned long long* bits) {
  const int p = blockIdx.x * kT + threadIdx.x;
  if (p >= np || pt_const[p]) return;
  const int ps = pstart[p], n = pstart[p + 1] - ps;
  for (int i = 1; i < n; ++i) {
    const int si = nat[blk_cam[ps + i]];
    if (si < 0) continue;
    for (int j = 0; j < i; ++j) {
      const int sj = nat[blk_cam[ps + j]];
      if (sj < 0 || sj == si) continue;
      unsigned long long* w1 = &bits[(size_t)si * words + (sj >> 6)];
      unsigned long long* w2 = &bits[(size_t)sj * words + (si >> 6)];
      const unsigned long long b1 = 1ull << (sj & 63), b2 = 1ull << (si & 63);
      if (!(*w1 & b1)) atomicOr(w1, b1);
      if (!(*w2 & b2)) atomicOr(w2, b2);
    }
  }
}

// The merge of a landmark's sorted blocks into records, shared by the counting and the writing pass: per camera the k-th
// reprojection block and the k-th depth block form record k (host build(), "Phase B").  emit(cam, slot, flags, reproj source or
// -1, depth source or -1).
template <class Emit>
__device__ __forceinline__ void merge_blocks(int ps, int n, const int32_t* blk_cam, const int32_t* blk_key, const uint32_t* blk_src, Emit&& emit) {
  int it = 0;
  while (it < n) {
    const int cam = blk_cam[ps + it];
    int je = it;
    while (je < n && blk_cam[ps + je] == cam) ++je;
    int mid = it;
    while (mid < je && !(blk_src[ps + mid] & 0x80000000u)) ++mid;
    const int nr = mid - it, nd = je - mid;
    const int key = blk_key[ps + it];
    const int slot = key == INT_MAX ? -1 : key;
    for (int k = 0; k < max(nr, nd); ++k) {
      uint32_t flags = 0;
      int64_t sr = -1, sd = -1;
      if (k < nr) { flags |= kRecHasReproj; sr = (int64_t)blk_src[ps + it + k]; }
      if (k < nd) { flags |= kRecHasDepth; sd = (int64_t)(blk_src[ps + mid + k] & 0x7fffffffu); }
      emit(cam, slot, flags, sr, sd);
    }
    it = je;
  }
}

__global__ __launch_bounds__(kT) void k_sortmerge(int np, const int32_t* pstart, int32_t* blk_cam, int32_t* blk_key, uint32_t* blk_src, const int32_t* slot_of_cam,
                                                  const uint8_t* pt_const, int dense_on, LmInfo* info, uint16_t* lm_slots) {
  const int p = blockIdx.x * kT + threadIdx.x;
  if (p >= np) return;
  const int ps = pstart[p], n = pstart[p + 1] - ps;
  LmInfo I{};
  if (n == 0) { info[p] = I; return; }
  for (int i = 0; i < n; ++i) { const int s = slot_of_cam[blk_cam[ps + i]]; blk_key[ps + i] = s < 0 ? INT_MAX : s; }
  // insertion sort by (key, camera, kind, source): `source` carries the kind in its top bit, so (kind, source) is one compare
  for (int i = 1; i < n; ++i) {
    const int k = blk_key[ps + i], c = blk_cam[ps + i];
    const uint32_t s = blk_src[ps + i];
    int j = i - 1;
    while (j >= 0) {
      const int kj = blk_key[ps + j], cj = blk_cam[ps + j];
      const uint32_t sj = blk_src[ps + j];
      const bool less = k != kj ? k < kj : (c != cj ? c < cj : s < sj);
      if (!less) break;
      blk_key[ps + j + 1] = kj; blk_cam[ps + j + 1] = cj; blk_src[ps + j + 1] = sj;
      --j;
    }
    blk_key[ps + j + 1] = k; blk_cam[ps + j + 1] = c; blk_src[ps + j + 1] = s;
  }
  const bool cpt = pt_const[p] != 0;
  int nrec = 0, nfix = 0, kv = 0, distinct = 0, last = -2;
  bool dup = false;
  uint64_t key[2] = {0, 0};
  int nkey = 0;
  merge_blocks(ps, n, blk_cam, blk_key, blk_src, [&](int, int slot, uint32_t, int64_t, int64_t) {
    if (slot < 0 && cpt) { ++nfix; return; }
    if (nkey < 6) {  // the first six record slots (16 bits each; constant cameras and slots beyond 65534 saturate)
      const uint64_t sk = slot >= 0 ? (uint64_t)min(slot, 0xfffe) : 0xffffull;
      key[nkey / 4] = (key[nkey / 4] << 16) | sk;
      ++nkey;
    }
    ++nrec;
    if (slot >= 0) {
      ++kv;
      if (slot != last) { if (distinct < kLmSlots) lm_slots[(size_t)p * kLmSlots + distinct] = (uint16_t)slot; ++distinct; last = slot; }
      else dup = true;
    }
  });
  for (; nkey < 6; ++nkey) key[nkey / 4] = (key[nkey / 4] << 16) | 0xffffull;
  key[1] = (key[1] << 32) | (uint64_t)(uint32_t)nrec;
  I.nrec = nrec; I.nfix = nfix; I.kv = kv; I.distinct = distinct; I.k1 = key[0]; I.k2 = key[1];
  if (nrec > kObsMax || distinct > kLocalCamsMax) I.flags |= LM_LONG;
  if (!dense_on || distinct > kDenseCams || (dup && !cpt)) I.flags |= LM_HEAVY;
  if (dup) I.flags |= LM_DUP;
  info[p] = I;
}

// candidates of the order: landmarks with records, ascending; and the landmarks that only fixed blocks reference
__global__ __launch_bounds__(kT) void k_flags(int np, const LmInfo* info, int32_t* has_rec, int32_t* fix_only, int32_t* nfix) {
  const int p = blockIdx.x * kT + threadIdx.x;
  if (p >= np) return;
  has_rec[p] = info[p].nrec > 0; fix_only[p] = info[p].nrec == 0 && info[p].nfix > 0; nfix[p] = info[p].nfix;
}
// Sort keys.  The host orders the landmarks by (first six record slots, record count, landmark); the same order from keys that hold
// each slot in `b` bits (b = bits of the slot count; a missing / constant camera becomes the all-ones value, above every slot) and the
// class (normal | heavy | long) on top, so that ONE stable radix sort over few bits replaces two 64-bit sorts and the class sort:
//   b <= 9 (up to 511 cameras):  key = class | s0 .. s5 | count  (2 + 6 b + 9 <= 65 - ... bits), one sort
//   otherwise: kB = s4 | s5 | count first, then kA = class | s0 .. s3 (stable)
__device__ __forceinline__ unsigned long long pack_slot(unsigned long long s16, int b) { const unsigned long long top = (1ull << b) - 1; return s16 >= top ? top : s16; }
__global__ __launch_bounds__(kT) void k_candidates(int np, const LmInfo* info, const int32_t* pos, int b, int single, unsigned long long* kA, unsigned long long* kB,
                                                   int32_t* cand, int32_t* counts) {
  const int p = blockIdx.x * kT + threadIdx.x;
  const bool valid = p < np && info[min(p, np - 1)].nrec > 0;
  LmInfo I{};
  if (valid) I = info[p];
  const unsigned long long cls = (I.flags & LM_LONG) ? 2ull : ((I.flags & LM_HEAVY) ? 1ull : 0ull);
  // class counts: one atomic per wave and class (150 k adds to one address take 1.8 ms)
  for (int c = 0; c < 3; ++c) {
    const unsigned long long m = __ballot(valid && cls == (unsigned long long)c);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(&counts[c], __popcll(m));
  }
  if (!valid) return;
  const int i = pos[p];
  unsigned long long sl[6];
#pragma unroll
  for (int q = 0; q < 4; ++q) sl[q] = pack_slot((I.k1 >> (16 * (3 - q))) & 0xffffull, b);
  sl[4] = pack_slot((I.k2 >> 48) & 0xffffull, b); sl[5] = pack_slot((I.k2 >> 32) & 0xffffull, b);
  const unsigned long long cnt9 = (unsigned long long)min(I.nrec, 511);  // chunked landmarks hold at most kObsMax records; longer ones are in class 2,
                                                                         // whose internal order the host phases redo anyway
  if (single) {
    unsigned long long k = cls;
#pragma unroll
    for (int q = 0; q < 6; ++q) k = (k << b) | sl[q];
    kA[i] = (k << 9) | cnt9;
  } else {
    kB[i] = (((sl[4] << b) | sl[5]) << 9) | cnt9;
    kA[i] = (((((cls << b) | sl[0]) << b | sl[1]) << b | sl[2]) << b) | sl[3];
  }
  cand[i] = p;
}
__global__ __launch_bounds__(kT) void k_gather_keys(int n, const int32_t* cand_sorted, const int32_t* pos, const unsigned long long* kA, unsigned long long* kA_g) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) kA_g[i] = kA[pos[cand_sorted[i]]];
}
__global__ __launch_bounds__(kT) void k_append_fixed_only(int np, const int32_t* fix_only, const int32_t* fpos, int base, int32_t* order) {
  const int p = blockIdx.x * kT + threadIdx.x;
  if (p < np && fix_only[p]) order[base + fpos[p]] = p;
}
__global__ __launch_bounds__(kT) void k_inverse(int n, const int32_t* order, const LmInfo* info, int32_t* inv, int32_t* nrec_k, int n_withrec) {
  const int k = blockIdx.x * kT + threadIdx.x;
  if (k >= n) return;
  const int p = order[k];
  inv[p] = k;
  nrec_k[k] = k < n_withrec ? info[p].nrec : 0;
}

// ---- the greedy chunk cut: one wave per segment ---------------------------------------------------------------------------
struct TmpChunk { int32_t first, npt, ncam, cam0; };  // segment-local camera offset
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// W = words of a camera set that can hold a bit (slots / 64, rounded up): lanes >= W never hold one.
__global__ __launch_bounds__(64) void k_cut(int np_chunked, int nseg, int W, const int32_t* order, const LmInfo* info, const uint16_t* lm_slots, const int32_t* pstart,
                                            const int32_t* blk_key, const int32_t* rec_off, TmpChunk* tmp_chunks, int32_t* tmp_cams, int32_t* lm_chunk,
                                            int32_t* seg_nchunks, int32_t* seg_ncams, int rec_cap) {
  const int sidx = blockIdx.x, lane = threadIdx.x;
  const int64_t k0 = (int64_t)np_chunked * sidx / nseg, k1 = (int64_t)np_chunked * (sidx + 1) / nseg;
  extern __shared__ unsigned long long s_mask[];  // [64 landmarks of a batch][W]: their camera sets, built by one lane each
  __shared__ int32_t s_ns[64], s_rp[64], s_hv[64], s_p[64];
  unsigned long long cur = 0;  // this lane's 64 slots of the open chunk's camera set
  int64_t c_first = k0;
  int c_nrec = 0, nchunks = 0, ncams = 0;
  int c_hv = 0;
  // number of set bits of a set spread over the lanes: only the lanes below W can hold any, and mostly a handful do
  auto set_size = [&](unsigned long long w) {
    unsigned long long live = __ballot(w != 0);
    int n = 0;
    while (live) {
      const int l = __builtin_ctzll(live);
      live &= live - 1;
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)w, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w >> 32), l);
      n += __popc(lo) + __popc(hi);
    }
    return n;
  };
  auto close_chunk = [&](int64_t end_pt) {
    if (end_pt == c_first) return;
    const int mine = __popcll(cur);
    int incl = mine;  // inclusive scan over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    const int total = __shfl(incl, 63, 64);
    int at = ncams + incl - mine;  // segment-local
    unsigned long long m = cur;
    int32_t* out = tmp_cams + rec_off[k0];  // capacity: a chunk has at most as many cameras as records
    while (m) { const int b = __builtin_ctzll(m); m &= m - 1; out[at++] = lane * 64 + b; }
    if (lane == 0) tmp_chunks[k0 + nchunks] = TmpChunk{(int32_t)c_first, (int32_t)(end_pt - c_first), total, ncams};
    ncams += total; ++nchunks;
    cur = 0; c_first = end_pt; c_nrec = 0;
  };
  for (int64_t base = k0; base < k1; base += 64) {
    const int64_t k = base + lane;
    __syncthreads();
    if (k < k1) {
      const int p = order[k];
      const LmInfo I = info[p];
      s_p[lane] = p; s_ns[lane] = I.distinct; s_rp[lane] = I.nrec; s_hv[lane] = (I.flags & LM_HEAVY) ? 1 : 0;
      unsigned long long* row = s_mask + (size_t)lane * W;
      for (int w = 0; w < W; ++w) row[w] = 0;
      if (I.distinct <= kLmSlots) {
        const uint16_t* sl = lm_slots + (size_t)p * kLmSlots;
        for (int t = 0; t < I.distinct; ++t) { const int sv = sl[t]; row[sv >> 6] |= 1ull << (sv & 63); }
      } else {  // a long camera list: from the sorted block keys
        const int ps = pstart[p], n = pstart[p + 1] - ps;
        for (int t = 0; t < n; ++t) { const int sv = blk_key[ps + t]; if (sv != INT_MAX) row[sv >> 6] |= 1ull << (sv & 63); }
      }
    }
    __syncthreads();
    const int nb = (int)min<int64_t>(64, k1 - base);
    for (int j = 0; j < nb; ++j) {
      const int64_t kk = base + j;
      const int rp = s_rp[j], hv = s_hv[j];
      const unsigned long long m = lane < W ? s_mask[(size_t)j * W + lane] : 0ull;  // the landmark's camera set
      const bool subset = __ballot((m & ~cur) != 0) == 0;
      const int nuni = subset ? -1 : set_size(cur | m);  // a subset does not grow the set (its size was admissible when it was formed)
      const bool first_of_chunk = kk == c_first;
      if (first_of_chunk) c_hv = hv;
      const bool too_big = (c_nrec + rp > (hv ? kObsMax : rec_cap)) || (kk - c_first + 1 > (hv ? kPtsMax : kDensePts)) || (nuni > (hv ? kLocalCamsMax : kDenseCams)) || (hv != c_hv);
      if (!first_of_chunk && too_big) {
        close_chunk(kk);
        cur = m; c_hv = hv;
      } else if (!subset) {
        cur |= m;
      }
      c_nrec += rp;
      if (lane == 0) lm_chunk[kk] = nchunks;
    }
  }
  close_chunk(k1);
  if (lane == 0) { seg_nchunks[sidx] = nchunks; seg_ncams[sidx] = ncams; }
}

// ---- the chunk cut for camera sets of up to 1024 slots (W <= 16 words), without the sequential walk over the landmarks -----------
// The greedy cut is a chain: a chunk that starts at landmark k ends at next(k), which depends on the landmarks from k on only.  So
// next(k) is computed for EVERY k in parallel (one thread walks the ~50 landmarks of the chunk that would start there, the camera
// set in W registers), and the segment's chunks are the chain k0 -> next(k0) -> ... followed by one thread per segment (~75 hops).
template <int W>
__global__ __launch_bounds__(kT) void k_lm_masks(int np_chunked, const int32_t* order, const LmInfo* info, const uint16_t* lm_slots, const int32_t* pstart,
                                                 const int32_t* blk_key, unsigned long long* mask, int32_t* rp, uint8_t* hv) {
  const int k = blockIdx.x * kT + threadIdx.x;
  if (k >= np_chunked) return;
  const int p = order[k];
  const LmInfo I = info[p];
  unsigned long long m[W];
#pragma unroll
  for (int w = 0; w < W; ++w) m[w] = 0;
  auto set = [&](int sv) {
#pragma unroll
    for (int w = 0; w < W; ++w) if ((sv >> 6) == w) m[w] |= 1ull << (sv & 63);
  };
  if (I.distinct <= kLmSlots) { for (int t = 0; t < I.distinct; ++t) set(lm_slots[(size_t)p * kLmSlots + t]); }
  else { const int ps = pstart[p], n = pstart[p + 1] - ps; for (int t = 0; t < n; ++t) { const int sv = blk_key[ps + t]; if (sv != INT_MAX) set(sv); } }
#pragma unroll
  for (int w = 0; w < W; ++w) mask[(size_t)k * W + w] = m[w];
  rp[k] = I.nrec; hv[k] = (I.flags & LM_HEAVY) ? 1 : 0;
}
__device__ __forceinline__ int seg_of(int64_t k, int np_chunked, int nseg) {
  int s = (int)min<int64_t>((k * nseg) / max(np_chunked, 1), nseg - 1);
  while (s > 0 && (int64_t)np_chunked * s / nseg > k) --s;
  while (s + 1 < nseg && (int64_t)np_chunked * (s + 1) / nseg <= k) ++s;
  return s;
}
template <int W>
__global__ __launch_bounds__(kT) void k_next(int np_chunked, int nseg, const unsigned long long* mask, const int32_t* rp, const uint8_t* hv, int32_t* next, int rec_cap, int pts_by_cams) {
  const int k = blockIdx.x * kT + threadIdx.x;
  if (k >= np_chunked) return;
  const int64_t k1 = (int64_t)np_chunked * (seg_of(k, np_chunked, nseg) + 1) / nseg;
  unsigned long long cur[W];
#pragma unroll
  for (int w = 0; w < W; ++w) cur[w] = 0;
  int nrec = 0;
  const int hv0 = hv[k];
  int64_t kk = k;
  for (; kk < k1; ++kk) {
    unsigned long long m[W];
    bool grow = false;
    int nuni = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) { m[w] = mask[(size_t)kk * W + w]; grow = grow || (m[w] & ~cur[w]) != 0; nuni += __popcll(cur[w] | m[w]); }
    const int nset = nuni;  // the chunk's camera set with this landmark
    if (!grow) nuni = -1;
    const int r = rp[kk], h = hv[kk];
    const bool too_big = (nrec + r > (h ? kObsMax : rec_cap)) || (kk - k + 1 > (h ? kPtsMax : dense_pts_cap(nset, pts_by_cams))) || (nuni > (h ? kLocalCamsMax : kDenseCams)) || (h != hv0);
    if (kk > k && too_big) break;
#pragma unroll
    for (int w = 0; w < W; ++w) cur[w] |= m[w];
    nrec += r;
  }
  next[k] = (int32_t)kk;
}
__global__ __launch_bounds__(64) void k_walk(int np_chunked, int nseg, const int32_t* next, int32_t* starts, int32_t* seg_nchunks) {
  const int sidx = blockIdx.x * 64 + threadIdx.x;
  if (sidx >= nseg) return;
  const int64_t k0 = (int64_t)np_chunked * sidx / nseg, k1 = (int64_t)np_chunked * (sidx + 1) / nseg;
  int n = 0;
  for (int64_t c = k0; c < k1; c = next[c]) starts[k0 + n++] = (int32_t)c;
  seg_nchunks[sidx] = n;
}
// one thread per chunk: its camera set, header, and (second pass, after the scan of the camera counts) its sorted camera list
template <int W>
__global__ __launch_bounds__(kT) void k_chunk_sets(int nchunks, int np_chunked, int nseg, const int32_t* cbase, const int32_t* starts, const unsigned long long* mask,
                                                   const int32_t* rec_off, const uint8_t* hv, ChunkHdr* chunks, int32_t* chunk_ncam, unsigned long long* csets) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c >= nchunks) return;
  int s = 0;  // the segment of chunk c: cbase is ascending, at most 64 entries
  while (s + 1 < nseg && cbase[s + 1] <= c) ++s;
  const int i = c - cbase[s];
  const int64_t k0 = (int64_t)np_chunked * s / nseg, k1 = (int64_t)np_chunked * (s + 1) / nseg;
  const int first = starts[k0 + i], end = (c + 1 < cbase[s + 1]) ? starts[k0 + i + 1] : (int)k1;
  unsigned long long u[W];
#pragma unroll
  for (int w = 0; w < W; ++w) u[w] = 0;
  for (int k = first; k < end; ++k)
#pragma unroll
    for (int w = 0; w < W; ++w) u[w] |= mask[(size_t)k * W + w];
  int n = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) { n += __popcll(u[w]); csets[(size_t)c * W + w] = u[w]; }
  ChunkHdr H{};
  H.rec0 = rec_off[first]; H.nrec = rec_off[end] - rec_off[first]; H.pt0 = first; H.npt = end - first; H.ncam = n; H.dense = hv[first] ? 0 : 1;
  chunks[c] = H;
  chunk_ncam[c] = n;
}
template <int W>
__global__ __launch_bounds__(kT) void k_chunk_cams(int nchunks, const unsigned long long* csets, const int32_t* cam0, ChunkHdr* chunks, int32_t* chunk_cams) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c >= nchunks) return;
  int at = cam0[c];
  chunks[c].cam0 = at;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    unsigned long long m = csets[(size_t)c * W + w];
    while (m) { const int b = __builtin_ctzll(m); m &= m - 1; chunk_cams[at++] = w * 64 + b; }
  }
}

__global__ __launch_bounds__(64) void k_seg_scan(int nseg, const int32_t* seg_nchunks, const int32_t* seg_ncams, int32_t* cbase, int32_t* cambase) {
  if (threadIdx.x == 0) {
    int a = 0, b = 0;
    for (int s = 0; s < nseg; ++s) { cbase[s] = a; cambase[s] = b; a += seg_nchunks[s]; b += seg_ncams[s]; }
    cbase[nseg] = a; cambase[nseg] = b;
  }
}
__global__ __launch_bounds__(kT) void k_chunks_final(int np_chunked, int nseg, const TmpChunk* tmp_chunks, const int32_t* tmp_cams, const int32_t* rec_off,
                                                     const int32_t* order, const LmInfo* info, const int32_t* cbase, const int32_t* cambase, ChunkHdr* chunks,
                                                     int32_t* chunk_cams) {
  const int sidx = blockIdx.x;
  const int64_t k0 = (int64_t)np_chunked * sidx / nseg;
  const int nch = cbase[sidx + 1] - cbase[sidx];
  for (int c = threadIdx.x; c < nch; c += kT) {
    const TmpChunk T = tmp_chunks[k0 + c];
    ChunkHdr H{};
    H.rec0 = rec_off[T.first]; H.nrec = rec_off[T.first + T.npt] - rec_off[T.first];
    H.pt0 = T.first; H.npt = T.npt; H.cam0 = cambase[sidx] + T.cam0; H.ncam = T.ncam;
    H.dense = (info[order[T.first]].flags & LM_HEAVY) ? 0 : 1;
    chunks[cbase[sidx] + c] = H;
  }
  const int ncam = cambase[sidx + 1] - cambase[sidx];
  const int32_t* src = tmp_cams + rec_off[k0];
  for (int i = threadIdx.x; i < ncam; i += kT) chunk_cams[cambase[sidx] + i] = src[i];
}

// ---- the record arrays -------------------------------------------------------------------------------------------------------
struct RecOut {
  int32_t* rec_cam; int32_t* rec_pt; uint32_t* rec_meta; double* rec_xy; double* rec_d; double* rec_m; double* rec_a;
  int32_t* pt_rec_start; uint16_t* pt_kv;
  int32_t* fx_cam; int32_t* fx_pt; uint32_t* fx_meta; double* fx_xy; double* fx_d; double* fx_m; double* fx_a;
};
__global__ __launch_bounds__(kT) void k_records(int n_order, int np_chunked, int n_withrec, int nrec_total, const int32_t* order, const LmInfo* info,
                                                const int32_t* pstart, const int32_t* blk_cam, const int32_t* blk_key, const uint32_t* blk_src, const int32_t* rec_off,
                                                const int32_t* fix_off, int nchunks, const ChunkHdr* chunks,
                                                const int32_t* chunk_cams, const uint8_t* pt_const, const double* obs_xy, const double* dobs_depth,
                                                const double* dobs_mag, const double* dobs_par, const double* shift, RecOut O, unsigned long long* counters,
                                                int32_t* err) {
  const int k = blockIdx.x * kT + threadIdx.x;
  unsigned long long my_nblk = 0;
  int my_var = 0;
  if (k < n_order) {
  const int p = order[k];
  const LmInfo I = info[p];
  const int ps = pstart[p], n = pstart[p + 1] - ps;
  const bool cpt = pt_const[p] != 0;
  const bool chunked = k < np_chunked;
  int w = k < n_withrec ? rec_off[k] : nrec_total;
  O.pt_rec_start[k] = w;
  ChunkHdr H{};
  const int32_t* cams = nullptr;
  if (chunked) {
    int lo = 0, hi = nchunks;  // the chunk whose landmarks [pt0, pt0 + npt) hold k
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunks[mid].pt0 <= k) lo = mid; else hi = mid; }
    H = chunks[lo];
    cams = chunk_cams + H.cam0;
  }
  int f = fix_off[p];
  unsigned long long nblk = 0;
  merge_blocks(ps, n, blk_cam, blk_key, blk_src, [&](int cam, int slot, uint32_t flags, int64_t sr, int64_t sd) {
    double u = 0.0, v = 0.0, d = 1.0, m = 0.0, a = 1.0;
    if (sr >= 0) { u = obs_xy[2 * sr]; v = obs_xy[2 * sr + 1]; }
    if (sd >= 0) {
      double b = 0.0, s = 0.0;
      if (shift) { b = shift[2 * cam]; s = shift[2 * cam + 1]; }
      d = dobs_depth[sd] * exp(s) + b; m = dobs_mag[sd]; a = dobs_par[sd];
      if (!(d > 0.0)) atomicOr(err, 2);
      d = log(d);  // the residual is log Z - log d: the records carry log d
    }
    if (slot < 0 && cpt) {
      O.fx_cam[f] = cam; O.fx_pt[f] = k; O.fx_meta[f] = flags; O.fx_xy[2 * f] = u; O.fx_xy[2 * f + 1] = v; O.fx_d[f] = d; O.fx_m[f] = m; O.fx_a[f] = a;
      ++f;
      return;
    }
    uint32_t lcam = kLcamConst;
    if (slot >= 0 && chunked) {
      int lo = 0, hi = H.ncam;  // lower bound in the chunk's sorted camera list
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (cams[mid] < slot) lo = mid + 1; else hi = mid; }
      lcam = (uint32_t)lo;
    }
    O.rec_cam[w] = cam; O.rec_pt[w] = k;
    O.rec_meta[w] = lcam | ((uint32_t)(k - H.pt0) << 8) | flags;
    O.rec_xy[2 * (size_t)w] = u; O.rec_xy[2 * (size_t)w + 1] = v; O.rec_d[w] = d; O.rec_m[w] = m; O.rec_a[w] = a;
    nblk += ((flags & kRecHasReproj) ? 1 : 0) + ((flags & kRecHasDepth) ? 1 : 0);
    ++w;
  });
  O.pt_kv[k] = (!cpt && k < n_withrec) ? (uint16_t)I.kv : (uint16_t)0xffff;
  my_nblk = nblk; my_var = (!cpt && k < n_withrec) ? 1 : 0;
  }
  // totals: one atomic per wave
  unsigned long long sb = my_nblk;
  int sv = my_var;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sb += __shfl_xor(sb, off, 64); sv += __shfl_xor(sv, off, 64); }
  if ((threadIdx.x & 63) == 0) { if (sb) atomicAdd(&counters[0], sb); if (sv) atomicAdd(&counters[1], (unsigned long long)sv); }
}


// ---- reduction tables of the dense chunks' slabs (k_reduce_slabs) -------------------------------------------------------------
// one entry per (chunk, block of S it contributes to) and per (chunk, camera): key = destination (block index, or nsb + slot),
// value = position of the source in the slabs (units of 18 doubles).  A stable sort by destination keeps chunk order inside a
// destination — the fixed summation order of k_reduce_slabs — exactly as the host's counting sort does.
__global__ __launch_bounds__(kT) void k_slab_entry_counts(int n_dense, const ChunkHdr* chunks, int32_t* nent) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c > n_dense) return;
  nent[c] = c < n_dense ? chunks[c].ncam * (chunks[c].ncam + 1) / 2 + chunks[c].ncam : 0;
}
__global__ __launch_bounds__(64) void k_slab_entries(int n_dense, const ChunkHdr* chunks, const int32_t* chunk_cams, BlockSky sky, int64_t nsb, int ndst,
                                                     const int32_t* ent0, uint32_t* key, int32_t* val, int32_t* cnt, int32_t* err) {
  const int c = blockIdx.x;
  if (c >= n_dense) return;
  const ChunkHdr H = chunks[c];
  const int32_t* cams = chunk_cams + H.cam0;
  const int nb = H.ncam * (H.ncam + 1) / 2;
  for (int e = threadIdx.x; e < nb + H.ncam; e += 64) {
    uint32_t k;
    int32_t v;
    if (e < nb) {
      int cj = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while (cj * (cj + 1) / 2 > e) --cj;
      while ((cj + 1) * (cj + 2) / 2 <= e) ++cj;
      const int ci = e - cj * (cj + 1) / 2;
      const int lo = cams[ci], hi = cams[cj];
      int64_t b;
      if (sky.index) b = sky.index[(int64_t)hi * sky.ns + lo];
      else b = sky.start[hi] + (lo - sky.first[hi]);
      if (b >= nsb) { atomicOr(&err[1], 1); b = -1; }
      k = b < 0 ? (uint32_t)ndst : (uint32_t)b;  // two cameras of the chunk that share no landmark anywhere: S has no such block
      v = H.slab0 + 2 * e;
    } else {
      k = (uint32_t)(nsb + cams[e - nb]);
      v = H.slab0 + 2 * nb + (e - nb);
    }
    key[ent0[c] + e] = k;
    val[ent0[c] + e] = v;
    if (k < (uint32_t)ndst) atomicAdd(&cnt[k], 1);
  }
}
__global__ __launch_bounds__(kT) void k_slab_mark_diag(int ncv, const int32_t* diag_block, int64_t nsb, uint8_t* is_diag) {
  const int sl = blockIdx.x * kT + threadIdx.x;
  if (sl >= ncv) return;
  const int32_t b = diag_block[sl];
  if (b >= 0 && b < nsb) is_diag[b] = 1;
}
__global__ __launch_bounds__(kT) void k_slab_part_counts(int ndst, const int32_t* cnt, int32_t* nparts) {
  const int d = blockIdx.x * kT + threadIdx.x;
  if (d > ndst) return;
  nparts[d] = d < ndst ? (cnt[d] + 15) / 16 : 0;
}
__global__ __launch_bounds__(kT) void k_slab_parts(int ndst, int64_t nsb, const int32_t* cnt, const int32_t* start, const int32_t* pstart, const uint8_t* is_diag,
                                                   RedDest* out) {
  const int d = blockIdx.x * kT + threadIdx.x;
  if (d >= ndst) return;
  const int n = cnt[d], s0 = start[d];
  RedDest R;
  R.kind = d >= nsb ? 2 : (is_diag[d] ? 1 : 0);
  R.dst = d >= nsb ? (int32_t)(d - nsb) : d;
  for (int q = 0, o = pstart[d]; q < n; q += 16, ++o) {
    R.s0 = s0 + q; R.s1 = s0 + min(q + 16, n);
    out[o] = R;
  }
}
}  // namespace

struct DevBuilder::Impl {
  hipStream_t s = nullptr;
  int nc = 0, np = 0;
  int64_t n_obs = 0, n_dobs = 0, nblk = 0;
  std::vector<void*> blocks;  // everything to give back
  template <class T> T* alloc(size_t n) { T* p = (T*)cached_malloc(std::max<size_t>(n, 1) * sizeof(T)); if (p) blocks.push_back(p); return p; }
  // rocPRIM calls with their temporary storage.  The block stays with the builder until its destructor has synchronised the
  // stream: the caching allocator is process-wide, a block given back while the call is still queued could be handed to another
  // host thread's handle at once and written from ITS stream (the hazard of DESIGN.md section 4b).
  template <class F>
  int with_temp(F&& f) {
    size_t bytes = 0;
    if (f(nullptr, bytes) != hipSuccess) return dfail(MPSFM_EHIP, "rocPRIM size query failed");
    void* tmp = alloc<uint8_t>(std::max<size_t>(bytes, 16));
    if (!tmp) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
    return f(tmp, bytes) == hipSuccess ? 0 : dfail(MPSFM_EHIP, "rocPRIM call failed");
  }
  // raw input
  int32_t *obs_cam = nullptr, *obs_pt = nullptr, *dobs_cam = nullptr, *dobs_pt = nullptr;
  double *obs_xy = nullptr, *dobs_depth = nullptr, *dobs_mag = nullptr, *dobs_par = nullptr, *shift = nullptr;
  uint8_t* pt_const = nullptr;
  // grouping
  int32_t *cnt_pt = nullptr, *cnt_cam = nullptr, *pstart = nullptr, *fill = nullptr, *blk_cam = nullptr, *blk_key = nullptr, *err = nullptr;
  uint32_t* blk_src = nullptr;
  unsigned long long* bits = nullptr;
  int32_t* nat = nullptr;
};

DevBuilder::DevBuilder() : m(new Impl()) {}
DevBuilder::~DevBuilder() {
  if (m->s) (void)hipStreamSynchronize(m->s);  // the blocks go back to a process-wide cache
  for (void* p : m->blocks) cached_free(p);
  delete m;
}

int DevBuilder::stage1(const mpsfm_ba_problem* P, hipStream_t stream, const std::vector<int32_t>& nat_slot, int ncv_real, std::vector<double>& cam_counts,
                       std::vector<uint64_t>& graph_bits, int graph_words, int64_t* max_blocks_per_landmark) {
  Impl& M = *m;
  M.s = stream; M.nc = P->n_cams; M.np = P->n_pts; M.n_obs = P->n_obs; M.n_dobs = P->n_dobs; M.nblk = P->n_obs + P->n_dobs;
  if (M.nblk > (int64_t)INT_MAX / 2) return dfail(MPSFM_EUNSUPPORTED, "device build: more than 2^30 residual blocks");
  const size_t no = (size_t)M.n_obs, nd = (size_t)M.n_dobs, np = (size_t)M.np, nc = (size_t)M.nc;
  M.obs_cam = M.alloc<int32_t>(no); M.obs_pt = M.alloc<int32_t>(no); M.obs_xy = M.alloc<double>(2 * no);
  M.dobs_cam = M.alloc<int32_t>(nd); M.dobs_pt = M.alloc<int32_t>(nd); M.dobs_depth = M.alloc<double>(nd); M.dobs_mag = M.alloc<double>(nd);
  M.dobs_par = M.alloc<double>(nd); M.pt_const = M.alloc<uint8_t>(np); M.nat = M.alloc<int32_t>(nc);
  M.cnt_pt = M.alloc<int32_t>(np + 1); M.cnt_cam = M.alloc<int32_t>(nc); M.pstart = M.alloc<int32_t>(np + 1); M.fill = M.alloc<int32_t>(np + 1);
  M.blk_cam = M.alloc<int32_t>((size_t)M.nblk); M.blk_key = M.alloc<int32_t>((size_t)M.nblk); M.blk_src = M.alloc<uint32_t>((size_t)M.nblk);
  M.err = M.alloc<int32_t>(4);
  M.bits = M.alloc<unsigned long long>((size_t)std::max(ncv_real, 1) * (size_t)std::max(graph_words, 1));
  if (P->shift_logscale) M.shift = M.alloc<double>(2 * nc);
  for (void* p : M.blocks) if (!p) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  int rc = 0;
  if (no) {
    if ((rc = staged_upload(M.obs_cam, P->obs_cam, 4 * no))) return rc;
    if ((rc = staged_upload(M.obs_pt, P->obs_pt, 4 * no))) return rc;
    if ((rc = staged_upload(M.obs_xy, P->obs_xy, 16 * no))) return rc;
  }
  if (nd) {
    if ((rc = staged_upload(M.dobs_cam, P->dobs_cam, 4 * nd))) return rc;
    if ((rc = staged_upload(M.dobs_pt, P->dobs_pt, 4 * nd))) return rc;
    if ((rc = staged_upload(M.dobs_depth, P->dobs_depth, 8 * nd))) return rc;
    if ((rc = staged_upload(M.dobs_mag, P->dobs_magnitude, 8 * nd))) return rc;
    if ((rc = staged_upload(M.dobs_par, P->dobs_param, 8 * nd))) return rc;
  }
  if (np && (rc = staged_upload(M.pt_const, P->pt_const, np))) return rc;
  if (nc && (rc = staged_upload(M.nat, nat_slot.data(), 4 * nc))) return rc;
  if (M.shift && (rc = staged_upload(M.shift, P->shift_logscale, 16 * nc))) return rc;
  if ((rc = staged_drain())) return rc;
  DB_TRY(hipMemsetAsync(M.cnt_pt, 0, 4 * (np + 1), M.s));
  DB_TRY(hipMemsetAsync(M.cnt_cam, 0, 4 * std::max<size_t>(nc, 1), M.s));
  DB_TRY(hipMemsetAsync(M.fill, 0, 4 * (np + 1), M.s));
  DB_TRY(hipMemsetAsync(M.err, 0, 16, M.s));
  DB_TRY(hipMemsetAsync(M.bits, 0, 8 * (size_t)std::max(ncv_real, 1) * (size_t)std::max(graph_words, 1), M.s));
  const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (M.nblk + kT - 1) / kT));
  hipLaunchKernelGGL(k_count, dim3(grid), dim3(kT), (size_t)4 * std::max(M.nc, 1), M.s, M.n_obs, M.n_dobs, M.obs_cam, M.obs_pt, M.dobs_cam, M.dobs_pt, M.dobs_depth,
                     M.nc, M.np, M.cnt_pt, M.cnt_cam, M.err);
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, M.cnt_pt, M.pstart, 0, np + 1, rocprim::plus<int32_t>(), M.s); }))) return rc;
  hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(kT), 0, M.s, M.n_obs, M.n_dobs, M.obs_cam, M.obs_pt, M.dobs_cam, M.dobs_pt, M.pstart, M.fill, M.blk_cam, M.blk_src);
  if (ncv_real > 0 && np)
    hipLaunchKernelGGL(k_graph, dim3((unsigned)((np + kT - 1) / kT)), dim3(kT), 0, M.s, M.np, M.pstart, M.blk_cam, M.pt_const, M.nat, graph_words, M.bits);
  // to the host: block counts per camera, the graph, the error flags, the longest block list
  std::vector<int32_t> cc(std::max<size_t>(nc, 1)), er(4), cp(np + 1);
  DB_TRY(hipMemcpyAsync(cc.data(), M.cnt_cam, 4 * nc, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipMemcpyAsync(er.data(), M.err, 16, hipMemcpyDeviceToHost, M.s));
  graph_bits.assign((size_t)std::max(ncv_real, 0) * (size_t)graph_words, 0);
  if (!graph_bits.empty()) DB_TRY(hipMemcpyAsync(graph_bits.data(), M.bits, 8 * graph_bits.size(), hipMemcpyDeviceToHost, M.s));
  // the longest block list: a max-reduction over the counts (rocPRIM)
  int32_t* d_max = M.alloc<int32_t>(1);
  if (!d_max) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::reduce(t, b, M.cnt_pt, d_max, 0, np + 1, rocprim::maximum<int32_t>(), M.s); }))) return rc;
  int32_t mx = 0;
  DB_TRY(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipStreamSynchronize(M.s));
  if (er[0] & 1) return dfail(MPSFM_EINVAL, "depth prior must be positive");
  cam_counts.assign(nc + 1, 0.0);
  for (size_t i = 0; i < nc; ++i) cam_counts[i] = (double)cc[i];
  *max_blocks_per_landmark = mx;
  return 0;
}

int DevBuilder::stage2(const std::vector<int32_t>& slot_of_cam, bool dense_on, int rec_cap, int pts_by_cams, DevBuildOut& out) {
  Impl& M = *m;
  const bool tr = std::getenv("MPSFM_DEVBUILD_TRACE") != nullptr;  // diagnostics: synchronise and print after every step
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!tr) return;
    (void)hipStreamSynchronize(M.s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mpsfm_ba] device build: %-34s %8.3f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  const size_t np = (size_t)M.np, nc = (size_t)M.nc;
  int rc = 0;
  int32_t* d_slot = M.alloc<int32_t>(nc);
  LmInfo* info = M.alloc<LmInfo>(np);
  uint16_t* lm_slots = M.alloc<uint16_t>(np * kLmSlots + 8);
  int32_t *has_rec = M.alloc<int32_t>(np + 1), *fix_only = M.alloc<int32_t>(np + 1), *nfix = M.alloc<int32_t>(np + 1);
  int32_t *pos = M.alloc<int32_t>(np + 1), *fpos = M.alloc<int32_t>(np + 1), *fix_off = M.alloc<int32_t>(np + 1);
  if (!d_slot || !info || !lm_slots || !has_rec || !fix_only || !nfix || !pos || !fpos || !fix_off) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  if (nc && (rc = staged_upload(d_slot, slot_of_cam.data(), 4 * nc))) return rc;
  if ((rc = staged_drain())) return rc;
  const unsigned gnp = (unsigned)std::max<size_t>(1, (np + kT - 1) / kT);
  if (np) hipLaunchKernelGGL(k_sortmerge, dim3(gnp), dim3(kT), 0, M.s, M.np, M.pstart, M.blk_cam, M.blk_key, M.blk_src, d_slot, M.pt_const, dense_on ? 1 : 0, info, lm_slots);
  lap("allocations + sort/merge per landmark");
  if (np) hipLaunchKernelGGL(k_flags, dim3(gnp), dim3(kT), 0, M.s, M.np, info, has_rec, fix_only, nfix);
  DB_TRY(hipMemsetAsync(has_rec + np, 0, 4, M.s));
  DB_TRY(hipMemsetAsync(fix_only + np, 0, 4, M.s));
  DB_TRY(hipMemsetAsync(nfix + np, 0, 4, M.s));
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, has_rec, pos, 0, np + 1, rocprim::plus<int32_t>(), M.s); }))) return rc;
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, fix_only, fpos, 0, np + 1, rocprim::plus<int32_t>(), M.s); }))) return rc;
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, nfix, fix_off, 0, np + 1, rocprim::plus<int32_t>(), M.s); }))) return rc;
  int32_t tot[3] = {0, 0, 0};
  DB_TRY(hipMemcpyAsync(&tot[0], pos + np, 4, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipMemcpyAsync(&tot[1], fpos + np, 4, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipMemcpyAsync(&tot[2], fix_off + np, 4, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipStreamSynchronize(M.s));
  lap("flags + three scans + sync");
  const int n_withrec = tot[0], n_fixonly = tot[1], n_fixed = tot[2];
  const int n_order = n_withrec + n_fixonly;
  out.np = n_order; out.nfixed = n_fixed;
  // ---- landmark order (see k_candidates)
  const size_t nw = (size_t)std::max(n_withrec, 1);
  unsigned long long *kA = M.alloc<unsigned long long>(nw), *kB = M.alloc<unsigned long long>(nw), *kS = M.alloc<unsigned long long>(nw), *kG = M.alloc<unsigned long long>(nw);
  int32_t *cand = M.alloc<int32_t>(nw), *cand1 = M.alloc<int32_t>(nw);
  int32_t* order = M.alloc<int32_t>((size_t)std::max(n_order, 1));
  int32_t* counts = M.alloc<int32_t>(4);
  if (!kA || !kB || !kS || !kG || !cand || !cand1 || !order || !counts) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  DB_TRY(hipMemsetAsync(counts, 0, 16, M.s));
  int32_t cl[4] = {0, 0, 0, 0};
  if (n_withrec > 0) {
    int nslots = 0;
    for (int32_t v : slot_of_cam) nslots = std::max(nslots, v + 1);
    int b = 1;
    while ((1 << b) < nslots + 1) ++b;  // every slot below the all-ones value
    const int single = 2 + 6 * b + 9 <= 64 ? 1 : 0;
    const unsigned gw = (unsigned)((n_withrec + kT - 1) / kT);
    hipLaunchKernelGGL(k_candidates, dim3(gnp), dim3(kT), 0, M.s, M.np, info, pos, b, single, kA, kB, cand, counts);
    if (single) {
      if ((rc = M.with_temp([&](void* t, size_t& sz) { return rocprim::radix_sort_pairs(t, sz, kA, kS, cand, order, (size_t)n_withrec, 0, (unsigned)(2 + 6 * b + 9), M.s); }))) return rc;
    } else {
      if ((rc = M.with_temp([&](void* t, size_t& sz) { return rocprim::radix_sort_pairs(t, sz, kB, kS, cand, cand1, (size_t)n_withrec, 0, (unsigned)(2 * b + 9), M.s); }))) return rc;
      hipLaunchKernelGGL(k_gather_keys, dim3(gw), dim3(kT), 0, M.s, n_withrec, cand1, pos, kA, kG);
      if ((rc = M.with_temp([&](void* t, size_t& sz) { return rocprim::radix_sort_pairs(t, sz, kG, kS, cand1, order, (size_t)n_withrec, 0, (unsigned)(2 + 4 * b), M.s); }))) return rc;
    }
    DB_TRY(hipMemcpyAsync(cl, counts, 16, hipMemcpyDeviceToHost, M.s));
  }
  lap("landmark order (radix sort)");
  if (n_fixonly > 0) hipLaunchKernelGGL(k_append_fixed_only, dim3(gnp), dim3(kT), 0, M.s, M.np, fix_only, fpos, n_withrec, order);
  int32_t* inv = M.alloc<int32_t>(np + 1);
  int32_t* nrec_k = M.alloc<int32_t>((size_t)n_order + 1);
  int32_t* rec_off = M.alloc<int32_t>((size_t)n_order + 2);
  if (!inv || !nrec_k || !rec_off) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  if (n_order > 0) hipLaunchKernelGGL(k_inverse, dim3((unsigned)((n_order + kT - 1) / kT)), dim3(kT), 0, M.s, n_order, order, info, inv, nrec_k, n_withrec);
  DB_TRY(hipMemsetAsync(nrec_k + n_order, 0, 4, M.s));
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, nrec_k, rec_off, 0, (size_t)n_order + 1, rocprim::plus<int32_t>(), M.s); }))) return rc;
  int32_t nrec_total = 0;
  DB_TRY(hipMemcpyAsync(&nrec_total, rec_off + n_order, 4, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipStreamSynchronize(M.s));
  lap("inverse + record offsets + sync");
  const int n_long = cl[2];
  const int np_chunked = n_withrec - n_long;
  out.np_chunked = np_chunked; out.n_long = n_long; out.nrec = nrec_total;
  if (n_long > 0) return MPSFM_DEVBUILD_FALLBACK;  // long tracks: the host build handles them
  // ---- chunk cut
  const int nseg = (int)std::max<int64_t>(1, std::min<int64_t>(64, np_chunked / 4096));
  int nslots = 0;
  for (int32_t v : slot_of_cam) nslots = std::max(nslots, v + 1);
  const int W = std::max(1, (nslots + 63) / 64);
  const bool jump = W <= 16;  // camera sets of up to 1024 slots: the parallel form (k_next / k_walk); beyond: one wave per segment (k_cut)
  int32_t *seg_nch = M.alloc<int32_t>(65), *seg_ncam = M.alloc<int32_t>(65), *cbase = M.alloc<int32_t>(66), *cambase = M.alloc<int32_t>(66);
  if (!seg_nch || !seg_ncam || !cbase || !cambase) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  int32_t hb[2][66];
  std::memset(hb, 0, sizeof(hb));
  TmpChunk* tmp_chunks = nullptr;
  int32_t *tmp_cams = nullptr, *starts = nullptr;
  unsigned long long* lmask = nullptr;
  uint8_t* hvk = nullptr;
  const int Wp = W <= 1 ? 1 : (W <= 2 ? 2 : (W <= 4 ? 4 : (W <= 8 ? 8 : 16)));  // the template instance
  if (np_chunked > 0 && jump) {
    lmask = M.alloc<unsigned long long>((size_t)np_chunked * Wp);
    int32_t* rpk = M.alloc<int32_t>((size_t)np_chunked);
    hvk = M.alloc<uint8_t>((size_t)np_chunked);
    int32_t* next = M.alloc<int32_t>((size_t)np_chunked);
    starts = M.alloc<int32_t>((size_t)np_chunked);
    if (!lmask || !rpk || !hvk || !next || !starts) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
    const unsigned gk = (unsigned)((np_chunked + kT - 1) / kT);
    DB_TRY(hipMemsetAsync(seg_ncam, 0, 4 * 65, M.s));
    switch (Wp) {
      case 1: hipLaunchKernelGGL(k_lm_masks<1>, dim3(gk), dim3(kT), 0, M.s, np_chunked, order, info, lm_slots, M.pstart, M.blk_key, lmask, rpk, hvk);
              hipLaunchKernelGGL(k_next<1>, dim3(gk), dim3(kT), 0, M.s, np_chunked, nseg, lmask, rpk, hvk, next, rec_cap, pts_by_cams); break;
      case 2: hipLaunchKernelGGL(k_lm_masks<2>, dim3(gk), dim3(kT), 0, M.s, np_chunked, order, info, lm_slots, M.pstart, M.blk_key, lmask, rpk, hvk);
              hipLaunchKernelGGL(k_next<2>, dim3(gk), dim3(kT), 0, M.s, np_chunked, nseg, lmask, rpk, hvk, next, rec_cap, pts_by_cams); break;
      case 4: hipLaunchKernelGGL(k_lm_masks<4>, dim3(gk), dim3(kT), 0, M.s, np_chunked, order, info, lm_slots, M.pstart, M.blk_key, lmask, rpk, hvk);
              hipLaunchKernelGGL(k_next<4>, dim3(gk), dim3(kT), 0, M.s, np_chunked, nseg, lmask, rpk, hvk, next, rec_cap, pts_by_cams); break;
      case 8: hipLaunchKernelGGL(k_lm_masks<8>, dim3(gk), dim3(kT), 0, M.s, np_chunked, order, info, lm_slots, M.pstart, M.blk_key, lmask, rpk, hvk);
              hipLaunchKernelGGL(k_next<8>, dim3(gk), dim3(kT), 0, M.s, np_chunked, nseg, lmask, rpk, hvk, next, rec_cap, pts_by_cams); break;
      default: hipLaunchKernelGGL(k_lm_masks<16>, dim3(gk), dim3(kT), 0, M.s, np_chunked, order, info, lm_slots, M.pstart, M.blk_key, lmask, rpk, hvk);
               hipLaunchKernelGGL(k_next<16>, dim3(gk), dim3(kT), 0, M.s, np_chunked, nseg, lmask, rpk, hvk, next, rec_cap, pts_by_cams); break;
    }
    hipLaunchKernelGGL(k_walk, dim3(1), dim3(64), 0, M.s, np_chunked, nseg, next, starts, seg_nch);
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(64), 0, M.s, nseg, seg_nch, seg_ncam, cbase, cambase);
    DB_TRY(hipMemcpyAsync(hb[0], cbase, 4 * (size_t)(nseg + 1), hipMemcpyDeviceToHost, M.s));
    DB_TRY(hipStreamSynchronize(M.s));
  } else if (np_chunked > 0) {
    tmp_chunks = M.alloc<TmpChunk>((size_t)np_chunked);
    tmp_cams = M.alloc<int32_t>((size_t)std::max(nrec_total, 1));
    int32_t* lm_chunk = M.alloc<int32_t>((size_t)np_chunked);
    if (!tmp_chunks || !tmp_cams || !lm_chunk) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
    hipLaunchKernelGGL(k_cut, dim3(nseg), dim3(64), (size_t)64 * W * 8, M.s, np_chunked, nseg, W, order, info, lm_slots, M.pstart, M.blk_key, rec_off, tmp_chunks, tmp_cams, lm_chunk, seg_nch, seg_ncam, rec_cap);
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(64), 0, M.s, nseg, seg_nch, seg_ncam, cbase, cambase);
    DB_TRY(hipMemcpyAsync(hb[0], cbase, 4 * (size_t)(nseg + 1), hipMemcpyDeviceToHost, M.s));
    DB_TRY(hipMemcpyAsync(hb[1], cambase, 4 * (size_t)(nseg + 1), hipMemcpyDeviceToHost, M.s));
    DB_TRY(hipStreamSynchronize(M.s));
  }
  lap("chunk cut");
  const int nchunks = hb[0][nseg];
  int ncams = hb[1][nseg];
  // ---- final tables (owned by the caller from here on)
  auto own = [&](size_t bytes) { return cached_malloc(std::max<size_t>(bytes, 8)); };
  const size_t nr = (size_t)std::max(nrec_total, 1), nf = (size_t)std::max(n_fixed, 1), no = (size_t)std::max(n_order, 1);
  out.d_chunks = (ChunkHdr*)own(sizeof(ChunkHdr) * (size_t)std::max(nchunks, 1));
  if (!out.d_chunks) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  unsigned long long* csets = nullptr;
  int32_t* cam0 = nullptr;
  if (jump && nchunks > 0) {
    // the chunks' camera sets and headers, the scan of their camera counts; the lists themselves once their table is sized
    csets = M.alloc<unsigned long long>((size_t)nchunks * Wp);
    int32_t* cncam = M.alloc<int32_t>((size_t)nchunks + 1);
    cam0 = M.alloc<int32_t>((size_t)nchunks + 1);
    if (!csets || !cncam || !cam0) { (void)hipStreamSynchronize(M.s); out.release(); return dfail(MPSFM_ENOMEM, "hipMalloc failed"); }
    const unsigned gc = (unsigned)((nchunks + kT - 1) / kT);
    switch (Wp) {
      case 1: hipLaunchKernelGGL(k_chunk_sets<1>, dim3(gc), dim3(kT), 0, M.s, nchunks, np_chunked, nseg, cbase, starts, lmask, rec_off, hvk, out.d_chunks, cncam, csets); break;
      case 2: hipLaunchKernelGGL(k_chunk_sets<2>, dim3(gc), dim3(kT), 0, M.s, nchunks, np_chunked, nseg, cbase, starts, lmask, rec_off, hvk, out.d_chunks, cncam, csets); break;
      case 4: hipLaunchKernelGGL(k_chunk_sets<4>, dim3(gc), dim3(kT), 0, M.s, nchunks, np_chunked, nseg, cbase, starts, lmask, rec_off, hvk, out.d_chunks, cncam, csets); break;
      case 8: hipLaunchKernelGGL(k_chunk_sets<8>, dim3(gc), dim3(kT), 0, M.s, nchunks, np_chunked, nseg, cbase, starts, lmask, rec_off, hvk, out.d_chunks, cncam, csets); break;
      default: hipLaunchKernelGGL(k_chunk_sets<16>, dim3(gc), dim3(kT), 0, M.s, nchunks, np_chunked, nseg, cbase, starts, lmask, rec_off, hvk, out.d_chunks, cncam, csets); break;
    }
    DB_TRY(hipMemsetAsync(cncam + nchunks, 0, 4, M.s));
    if ((rc = M.with_temp([&](void* t, size_t& sz) { return rocprim::exclusive_scan(t, sz, cncam, cam0, 0, (size_t)nchunks + 1, rocprim::plus<int32_t>(), M.s); }))) { (void)hipStreamSynchronize(M.s); out.release(); return rc; }
    DB_TRY(hipMemcpyAsync(&ncams, cam0 + nchunks, 4, hipMemcpyDeviceToHost, M.s));
    DB_TRY(hipStreamSynchronize(M.s));
  }
  out.d_chunk_cams = (int32_t*)own(4 * (size_t)std::max(ncams, 1));
  out.d_rec_cam = (int32_t*)own(4 * nr); out.d_rec_pt = (int32_t*)own(4 * nr); out.d_rec_meta = (uint32_t*)own(4 * nr);
  out.d_rec_xy = (double*)own(16 * nr); out.d_rec_d = (double*)own(8 * nr); out.d_rec_m = (double*)own(8 * nr); out.d_rec_a = (double*)own(8 * nr);
  out.d_pt_rec_start = (int32_t*)own(4 * (no + 1)); out.d_pt_kv = (uint16_t*)own(2 * (no + 1));
  out.d_fx_cam = (int32_t*)own(4 * nf); out.d_fx_pt = (int32_t*)own(4 * nf); out.d_fx_meta = (uint32_t*)own(4 * nf);
  out.d_fx_xy = (double*)own(16 * nf); out.d_fx_d = (double*)own(8 * nf); out.d_fx_m = (double*)own(8 * nf); out.d_fx_a = (double*)own(8 * nf);
  unsigned long long* counters = M.alloc<unsigned long long>(2);
  void* all[] = {out.d_chunks, out.d_chunk_cams, out.d_rec_cam, out.d_rec_pt, out.d_rec_meta, out.d_rec_xy, out.d_rec_d, out.d_rec_m, out.d_rec_a, out.d_pt_rec_start,
                 out.d_pt_kv, out.d_fx_cam, out.d_fx_pt, out.d_fx_meta, out.d_fx_xy, out.d_fx_d, out.d_fx_m, out.d_fx_a, counters};
  for (void* p : all) if (!p) { (void)hipStreamSynchronize(M.s); out.release(); return dfail(MPSFM_ENOMEM, "hipMalloc failed"); }
  DB_TRY(hipMemsetAsync(counters, 0, 16, M.s));
  if (nchunks > 0 && jump) {
    const unsigned gc = (unsigned)((nchunks + kT - 1) / kT);
    switch (Wp) {
      case 1: hipLaunchKernelGGL(k_chunk_cams<1>, dim3(gc), dim3(kT), 0, M.s, nchunks, csets, cam0, out.d_chunks, out.d_chunk_cams); break;
      case 2: hipLaunchKernelGGL(k_chunk_cams<2>, dim3(gc), dim3(kT), 0, M.s, nchunks, csets, cam0, out.d_chunks, out.d_chunk_cams); break;
      case 4: hipLaunchKernelGGL(k_chunk_cams<4>, dim3(gc), dim3(kT), 0, M.s, nchunks, csets, cam0, out.d_chunks, out.d_chunk_cams); break;
      case 8: hipLaunchKernelGGL(k_chunk_cams<8>, dim3(gc), dim3(kT), 0, M.s, nchunks, csets, cam0, out.d_chunks, out.d_chunk_cams); break;
      default: hipLaunchKernelGGL(k_chunk_cams<16>, dim3(gc), dim3(kT), 0, M.s, nchunks, csets, cam0, out.d_chunks, out.d_chunk_cams); break;
    }
  } else if (nchunks > 0)
    hipLaunchKernelGGL(k_chunks_final, dim3(nseg), dim3(kT), 0, M.s, np_chunked, nseg, tmp_chunks, tmp_cams, rec_off, order, info, cbase, cambase, out.d_chunks, out.d_chunk_cams);
  RecOut O{out.d_rec_cam, out.d_rec_pt, out.d_rec_meta, out.d_rec_xy, out.d_rec_d, out.d_rec_m, out.d_rec_a, out.d_pt_rec_start, out.d_pt_kv,
           out.d_fx_cam, out.d_fx_pt, out.d_fx_meta, out.d_fx_xy, out.d_fx_d, out.d_fx_m, out.d_fx_a};
  if (n_order > 0)
    hipLaunchKernelGGL(k_records, dim3((unsigned)((n_order + kT - 1) / kT)), dim3(kT), 0, M.s, n_order, np_chunked, n_withrec, nrec_total, order, info, M.pstart, M.blk_cam,
                       M.blk_key, M.blk_src, rec_off, fix_off, nchunks, out.d_chunks, out.d_chunk_cams, M.pt_const, M.obs_xy, M.dobs_depth, M.dobs_mag,
                       M.dobs_par, M.shift, O, counters, M.err);
  lap("final tables: allocate, chunks, records");
  // the sentinel entry of pt_rec_start / pt_kv
  DB_TRY(hipMemcpyAsync(out.d_pt_rec_start + n_order, &nrec_total, 4, hipMemcpyHostToDevice, M.s));
  const uint16_t kv_none = 0xffff;
  DB_TRY(hipMemcpyAsync(out.d_pt_kv + n_order, &kv_none, 2, hipMemcpyHostToDevice, M.s));
  // ---- to the host: chunk headers, camera lists, the landmark order, counters
  out.chunks.resize((size_t)nchunks); out.chunk_cams.resize((size_t)ncams); out.order.resize((size_t)n_order);
  unsigned long long hc[2] = {0, 0};
  int32_t er[4] = {0, 0, 0, 0};
  if (nchunks) DB_TRY(hipMemcpyAsync(out.chunks.data(), out.d_chunks, sizeof(ChunkHdr) * (size_t)nchunks, hipMemcpyDeviceToHost, M.s));
  if (ncams) DB_TRY(hipMemcpyAsync(out.chunk_cams.data(), out.d_chunk_cams, 4 * (size_t)ncams, hipMemcpyDeviceToHost, M.s));
  if (n_order) DB_TRY(hipMemcpyAsync(out.order.data(), order, 4 * (size_t)n_order, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipMemcpyAsync(hc, counters, 16, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipMemcpyAsync(er, M.err, 16, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipStreamSynchronize(M.s));
  lap("downloads");
  if (er[0] & 2) { out.release(); return dfail(MPSFM_EINVAL, "shifted/scaled depth prior must be positive"); }
  out.nblk_reduced = (int64_t)hc[0]; out.nvarpts = (double)hc[1];
  return 0;
}

// Reduction tables of the dense chunks' slabs from the device copies of the chunk headers (slab offsets set) and camera lists.
int DevBuilder::slab_tables(const ChunkHdr* d_chunks, int n_dense, const int32_t* d_chunk_cams, const BlockSky& sky, int64_t nsb, int ncv,
                            const int32_t* d_diag_block, RedDest** d_dests, int32_t* n_dests, int32_t** d_srcs, int64_t* n_srcs) {
  Impl& M = *m;
  *d_dests = nullptr; *d_srcs = nullptr; *n_dests = 0; *n_srcs = 0;
  if (n_dense <= 0) return 0;
  const int ndst = (int)(nsb + ncv);
  int rc = 0;
  int32_t* nent = M.alloc<int32_t>((size_t)n_dense + 1);
  int32_t* ent0 = M.alloc<int32_t>((size_t)n_dense + 1);
  int32_t* cnt = M.alloc<int32_t>((size_t)ndst + 1);
  int32_t* start = M.alloc<int32_t>((size_t)ndst + 1);
  int32_t* nparts = M.alloc<int32_t>((size_t)ndst + 1);
  int32_t* pstart = M.alloc<int32_t>((size_t)ndst + 1);
  uint8_t* is_diag = M.alloc<uint8_t>((size_t)std::max<int64_t>(nsb, 1));
  if (!nent || !ent0 || !cnt || !start || !nparts || !pstart || !is_diag) return dfail(MPSFM_ENOMEM, "hipMalloc failed");
  hipLaunchKernelGGL(k_slab_entry_counts, dim3((unsigned)((n_dense + 1 + kT - 1) / kT)), dim3(kT), 0, M.s, n_dense, d_chunks, nent);
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, nent, ent0, 0, (size_t)n_dense + 1, rocprim::plus<int32_t>(), M.s); }))) return rc;
  int32_t n_ent = 0;
  DB_TRY(hipMemcpyAsync(&n_ent, ent0 + n_dense, 4, hipMemcpyDeviceToHost, M.s));
  DB_TRY(hipMemsetAsync(cnt, 0, 4 * ((size_t)ndst + 1), M.s));
  DB_TRY(hipMemsetAsync(is_diag, 0, (size_t)std::max<int64_t>(nsb, 1), M.s));
  DB_TRY(hipMemsetAsync(M.err, 0, 16, M.s));
  DB_TRY(hipStreamSynchronize(M.s));
  if (n_ent <= 0) return 0;
  uint32_t* key = M.alloc<uint32_t>((size_t)n_ent);
  uint32_t* key_s = M.alloc<uint32_t>((size_t)n_ent);
  int32_t* val = M.alloc<int32_t>((size_t)n_ent);
  int32_t* val_s = (int32_t*)cached_malloc(4 * (size_t)n_ent);  // becomes the source list: owned by the caller
  if (!key || !key_s || !val || !val_s) { cached_free(val_s); return dfail(MPSFM_ENOMEM, "hipMalloc failed"); }
  auto bail = [&](int code) { (void)hipStreamSynchronize(M.s); cached_free(val_s); return code; };
  hipLaunchKernelGGL(k_slab_entries, dim3((unsigned)n_dense), dim3(64), 0, M.s, n_dense, d_chunks, d_chunk_cams, sky, nsb, ndst, ent0, key, val, cnt, M.err);
  hipLaunchKernelGGL(k_slab_mark_diag, dim3((unsigned)((ncv + kT - 1) / kT)), dim3(kT), 0, M.s, ncv, d_diag_block, nsb, is_diag);
  unsigned bits = 1;
  while ((1u << bits) <= (unsigned)ndst) ++bits;  // keys 0 .. ndst (ndst: no destination)
  if ((rc = M.with_temp([&](void* t, size_t& sz) { return rocprim::radix_sort_pairs(t, sz, key, key_s, val, val_s, (size_t)n_ent, 0, bits, M.s); }))) return bail(rc);
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, cnt, start, 0, (size_t)ndst + 1, rocprim::plus<int32_t>(), M.s); }))) return bail(rc);
  hipLaunchKernelGGL(k_slab_part_counts, dim3((unsigned)((ndst + 1 + kT - 1) / kT)), dim3(kT), 0, M.s, ndst, cnt, nparts);
  if ((rc = M.with_temp([&](void* t, size_t& b) { return rocprim::exclusive_scan(t, b, nparts, pstart, 0, (size_t)ndst + 1, rocprim::plus<int32_t>(), M.s); }))) return bail(rc);
  int32_t tot[2] = {0, 0}, er[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(&tot[0], start + ndst, 4, hipMemcpyDeviceToHost, M.s) != hipSuccess || hipMemcpyAsync(&tot[1], pstart + ndst, 4, hipMemcpyDeviceToHost, M.s) != hipSuccess ||
      hipMemcpyAsync(er, M.err, 16, hipMemcpyDeviceToHost, M.s) != hipSuccess || hipStreamSynchronize(M.s) != hipSuccess)
    return bail(dfail(MPSFM_EHIP, "device build: copying the slab table sizes failed"));
  if (er[1]) return bail(dfail(MPSFM_EUNSUPPORTED, "internal: block index beyond S"));
  RedDest* dests = (RedDest*)cached_malloc(sizeof(RedDest) * (size_t)std::max(tot[1], 1));
  if (!dests) return bail(dfail(MPSFM_ENOMEM, "hipMalloc failed"));
  hipLaunchKernelGGL(k_slab_parts, dim3((unsigned)((ndst + kT - 1) / kT)), dim3(kT), 0, M.s, ndst, nsb, cnt, start, pstart, is_diag, dests);
  if (hipGetLastError() != hipSuccess) { cached_free(dests); return bail(dfail(MPSFM_EHIP, "device build: slab table kernels failed")); }
  *d_dests = dests; *n_dests = tot[1]; *d_srcs = val_s; *n_srcs = tot[0];
  return 0;
}

void DevBuildOut::release() {
  void* all[] = {d_chunks, d_chunk_cams, d_rec_cam, d_rec_pt, d_rec_meta, d_rec_xy, d_rec_d, d_rec_m, d_rec_a, d_pt_rec_start, d_pt_kv,
                 d_fx_cam, d_fx_pt, d_fx_meta, d_fx_xy, d_fx_d, d_fx_m, d_fx_a};
  for (void* p : all) cached_free(p);
  d_chunks = nullptr; d_chunk_cams = nullptr; d_rec_cam = d_rec_pt = nullptr; d_rec_meta = nullptr; d_rec_xy = d_rec_d = d_rec_m = d_rec_a = nullptr;
  d_pt_rec_start = nullptr; d_pt_kv = nullptr; d_fx_cam = d_fx_pt = nullptr; d_fx_meta = nullptr; d_fx_xy = d_fx_d = d_fx_m = d_fx_a = nullptr;
}

}  // namespace mpsfm
