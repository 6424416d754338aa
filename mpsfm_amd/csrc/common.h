// Shared device/host definitions for libmpsfm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <math.h>

#include "../../include/mpsfm_hip.h"
#include "chol_plan.h"

namespace mpsfm {

// tuning knobs (overridable with -D for experiments; scripts/sweep_variants.sh)
#ifndef MPSFM_TILE_CAMS
#define MPSFM_TILE_CAMS 16
#endif
#ifndef MPSFM_ITEM_PAIRS
#define MPSFM_ITEM_PAIRS 64
#endif
#ifndef MPSFM_CAM_COPIES
#define MPSFM_CAM_COPIES 3  // LDS copies of the per-camera accumulators of the track sweep
#endif
#ifndef MPSFM_ENT_STAGE
#define MPSFM_ENT_STAGE 1024
#endif
#ifndef MPSFM_OBS_MAX
#define MPSFM_OBS_MAX 252  // 252 records x 144 B of Z + the accumulators = 53.4 KB of LDS: three workgroups of the dense sweep per CU
#endif

// ---- track-sweep chunk geometry ----------------------------------------------------------
// A chunk = a group of consecutive (re-ordered) landmarks processed by one workgroup.
typedef double v4d __attribute__((ext_vector_type(4)));  // accumulators of v_mfma_f64_16x16x4_f64
constexpr int kThreads = 256;   // workgroup size of the sweep kernels
constexpr int kObsMax = MPSFM_OBS_MAX;    // merged (camera, landmark) records per chunk
constexpr int kPtsMax = MPSFM_OBS_MAX / 2;     // landmarks per chunk
constexpr int kTileCams = MPSFM_TILE_CAMS;   // local cameras whose U blocks / g_c are accumulated in LDS
#ifndef MPSFM_LOCAL_CAMS
#define MPSFM_LOCAL_CAMS 254
#endif
constexpr int kLocalCamsMax = MPSFM_LOCAL_CAMS;  // local camera list length (beyond kTileCams: direct atomics); < 255 (8-bit local index)
static_assert(kLocalCamsMax < 255, "local camera indices are 8 bits, 0xff marks a constant camera");
constexpr int kPairGroup = 6;   // lanes cooperating on one 6x6 block of the reduced system (one row each)
constexpr int kItemPairs = MPSFM_ITEM_PAIRS;  // pairs per Schur work item (heavier blocks are split for balance)
constexpr int kEntStage = MPSFM_ENT_STAGE; // pair entries of a chunk staged in LDS (larger chunks read them from HBM)
static_assert(kObsMax <= 256 && kPtsMax <= 128, "chunk-relative record / landmark indices are packed into 8 bits (rec_meta, pair entries)");
constexpr int kMaxDevices = 16;  // per-device pools (allocator cache, streams, staging buffers) are arrays of this length
constexpr int kWStride = 18;    // row stride (doubles) of the per-record W block in LDS: 144 B, keeps rows 16-B aligned for ds_read_b128
constexpr int kCamRec = 24;     // doubles per camera table record

// camera table record: R[9] t[3] K[4] cs[6] pad[2]
// cs = Jacobi column scale * tangent mask (0 for constant cameras / fixed coordinates)

struct ChunkHdr {
  int32_t rec0, nrec;    // merged records [rec0, rec0+nrec)
  int32_t pt0, npt;      // re-ordered landmarks [pt0, pt0+npt)
  int32_t cam0, ncam;    // local camera list in chunk_cams[cam0 ..)
  int32_t blk0, nblk;    // Schur work items (a destination block of S + a run of its pairs) of this chunk
  int32_t ent0, nent;    // the chunk's pair entries
  int32_t dense;         // 1: at most kDenseCams local cameras, kDensePts landmarks and one record per (camera, variable landmark): swept by
                         //    k_track_sweep_dense (Schur products as one small dense product on the matrix pipe, results through the chunk's
                         //    slab); such chunks have no pair tables
  int32_t slab0;         // dense chunks: start of the chunk's slab in SweepArgs::slab, in units of 18 doubles
};
constexpr int kDenseCams = 16;  // 16 cameras x 6 rows = 96 rows = six 16-row MFMA tiles
constexpr int kDensePts = 96;   // landmarks of a dense chunk
// ... by the size of its camera set when `by_cams` (small problems: one workgroup per chunk and CU, the slowest chunk sets the pace and
// a chunk's Schur products grow with landmarks x tile pairs): 96 up to 6 cameras, 64 up to 8, 48 up to 12, 32 beyond
__host__ __device__ inline int dense_pts_cap(int ncams, int by_cams) { return (!by_cams || ncams <= 6) ? kDensePts : (ncams <= 8 ? 64 : (ncams <= 12 ? 48 : 32)); }
// slab of a dense chunk: [ncam (ncam + 1) / 2 blocks (ci <= cj at cj (cj + 1) / 2 + ci) x 36][ncam x (g_c 6 | W V^-1 g_p 6 | diag U 6)] doubles,
// written with plain stores by k_track_sweep_dense and summed per destination by k_reduce_slabs
__host__ __device__ inline int64_t slab_doubles(int ncam) { return (int64_t)(ncam * (ncam + 1) / 2) * 36 + (int64_t)ncam * 18; }

// One destination of the slab reduction: a 6x6 block of S (kind 0: all 36 entries, 1: a diagonal block, upper triangle) or a camera's
// vectors (kind 2).  Destinations with many sources are split into parts (every part adds atomically into the zeroed buffer).
struct RedDest { int32_t kind; int32_t dst; int32_t s0, s1; };

// A landmark whose track does not fit a chunk (more than kObsMax records or kLocalCamsMax cameras)
// is swept by a workgroup of its own that strides over the records (k_long_track_sweep).
struct LongHdr {
  int32_t rec0, nrec;    // its records (variable cameras first, sorted by slot)
  int32_t pt;            // re-ordered landmark index
  int32_t kv;            // records with a variable camera
  int64_t w0;            // first row of its W scratch (18 doubles per record)
};

// record meta word: lcam | lpt << 8 | flags << 16
constexpr uint32_t kRecHasReproj = 1u << 16;
constexpr uint32_t kRecHasDepth = 1u << 17;
constexpr uint32_t kLcamConst = 255;

struct LossParams {
  int32_t reproj_type;
  double reproj_a;
  double reproj_mag;
  int32_t depth_type;
};

// scalar slots of the reduced buffer tail (all-reduced with S)
constexpr int kMaxRankSlots = 64;  // ranks whose landmark-gradient maxima travel exactly (one slot each in the summed tail)
enum { SC_COST = 0, SC_BAD = 1, SC_GMAX_PTS = 2, SC_RANK0 = 8, SC_COUNT = 8 + kMaxRankSlots };

// scalar slots produced by the update sweep + camera update (host reads these each iteration)
enum {
  U_CAND_COST = 0, U_BAD = 1, U_MCC = 2, U_STEP_SQ_PTS = 3, U_XN_SQ_PTS = 4,
  U_STEP_SQ_CAMS = 5, U_XN_SQ_CAMS = 6, U_GMAX_CAMS = 7, U_X_COST = 8, U_X_BAD = 9,
  U_GMAX_PTS = 10, U_CHOL_FAIL = 11, U_COUNT = 16
};

// Levenberg-Marquardt control block on the device.  The accept / reject decision of an iteration (Ceres
// trust_region_minimizer.cc) is taken by a one-thread kernel (k_lm_decide) from the scalars the sweeps leave behind, so the
// host enqueues iteration i+1 while iteration i still runs and only reads a copy of this block one iteration late: no
// stream synchronisation inside the loop.  Every kernel of an iteration returns at once when `term` says the solve is
// over (the one iteration enqueued ahead of the news).
struct LmHead {  // what travels to the host every iteration
  double radius, decrease_factor, x_norm, cur_cost, fixed_cost, initial_cost;
  double last_x_cost, last_cand, last_rel, last_step_norm, last_mcc;  // diagnostics of the last decided iteration
  int32_t term;          // kLmRunning, MPSFM_TERM_* once decided, kLmNumericError: the initial point cannot be evaluated
  int32_t iter, invalid_run, check_gradient, accepted, n_success, n_unsuccess, n_cost_evals, n_jac_evals, last_chol_fail, trace_len, pad_;
};
struct LmCtl : LmHead {
  double trace_cost[MPSFM_MAX_TRACE], trace_radius[MPSFM_MAX_TRACE];
  uint8_t trace_accepted[MPSFM_MAX_TRACE];
};
constexpr int32_t kLmRunning = -1, kLmNumericError = -2;
struct LmOpts {
  double function_tolerance, gradient_tolerance, parameter_tolerance, min_relative_decrease, max_radius, min_radius;
  int32_t max_iterations, max_invalid_steps;
};
__device__ inline bool lm_over(const LmCtl* c) { return c != nullptr && __hip_atomic_load(&c->term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kLmRunning; }
__device__ inline double lm_radius_of(const LmCtl* c) { return __hip_atomic_load(&c->radius, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- small device math -------------------------------------------------------------------
// Reciprocal, square root and reciprocal square root in a handful of instructions: the hardware estimate (v_rcp_f64 / v_rsq_f64,
// ~2^-22 relative) refined by two Newton steps (error ~ e^4: below one ulp; the last bit may differ from the IEEE-rounded quotient
// the compiler's ~30-instruction expansion of `/` and sqrt() delivers).  The sweeps take five quotients and three roots per record.
__host__ __device__ inline double fast_rcp(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  return fma(y, fma(-x, y, 1.0), y);
#else
  return 1.0 / x;
#endif
}
// s = sqrt(x), r = 1 / sqrt(x) for x > 0 (x == 0: s = 0, r = inf)
__host__ __device__ inline void fast_sqrt_rsqrt(double x, double& s, double& r) {
#ifdef __HIP_DEVICE_COMPILE__
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double e = fma(-g, h, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  e = fma(-g, h, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  s = x == 0.0 ? 0.0 : g;
  r = x == 0.0 ? y : 2.0 * h;
#else
  s = sqrt(x); r = 1.0 / s;
#endif
}
__host__ __device__ inline double fast_sqrt(double x) { double s, r; fast_sqrt_rsqrt(x, s, r); return s; }

__host__ __device__ inline void loss_eval(int type, double a, double s, double& rho0, double& rho1) {
  if (type == MPSFM_LOSS_SOFT_L1) {
    const double b = a * a, c = fast_rcp(b);
    const double sum = 1.0 + s * c;
    double tmp, itmp;
    fast_sqrt_rsqrt(sum, tmp, itmp);
    rho0 = 2.0 * b * (tmp - 1.0);
    rho1 = fmax(DBL_MIN, itmp);
  } else if (type == MPSFM_LOSS_CAUCHY) {
    const double b = a * a, c = fast_rcp(b);
    const double sum = 1.0 + s * c;
    rho0 = b * log(sum);
    rho1 = fmax(DBL_MIN, fast_rcp(sum));
  } else {
    rho0 = s;
    rho1 = 1.0;
  }
}

__host__ __device__ inline void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// EigenQuaternionManifold::Plus, q' = exp(d) * q
__host__ __device__ inline void quat_plus(const double* q, const double* d, double* out) {
  const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0.0) { out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3]; return; }
  const double sbd = sin(n) / n;
  const double px = sbd * d[0], py = sbd * d[1], pz = sbd * d[2], pw = cos(n);
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  out[0] = pw * qx + px * qw + py * qz - pz * qy;
  out[1] = pw * qy - px * qz + py * qw + pz * qx;
  out[2] = pw * qz + px * qy - py * qx + pz * qw;
  out[3] = pw * qw - px * qx - py * qy - pz * qz;
}

// 3x3 SPD inverse via LL^T; V packed upper [00 01 02 11 12 22]
__host__ __device__ inline bool spd3_inverse(const double* V, double* Vi) {
  const double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5];
  if (!(a > 0.0)) return false;
  const double l00 = sqrt(a);
  const double l10 = b / l00, l20 = c / l00;
  const double t11 = d - l10 * l10;
  if (!(t11 > 0.0)) return false;
  const double l11 = sqrt(t11);
  const double l21 = (e - l20 * l10) / l11;
  const double t22 = f - l20 * l20 - l21 * l21;
  if (!(t22 > 0.0)) return false;
  const double l22 = sqrt(t22);
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  Vi[0] = i00 * i00 + i10 * i10 + i20 * i20;
  Vi[1] = i10 * i11 + i20 * i21;
  Vi[2] = i20 * i22;
  Vi[3] = i11 * i11 + i21 * i21;
  Vi[4] = i21 * i22;
  Vi[5] = i22 * i22;
  return true;
}

// The inverse Cholesky factor of a 3x3 SPD matrix: V = L L^T, F = L^-1 (lower), packed [00 10 11 20 21 22],
// so that V^-1 = F^T F.  With Z = W F^T:  W V^-1 W'^T = Z Z'^T  and  W V^-1 g = Z (F g).
__host__ __device__ inline bool spd3_inv_factor(const double* V, double* F) {
  const double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5];
  if (!(a > 0.0)) return false;
  double l00, i00, l11, i11, l22, i22;  // the factor's diagonal and its inverse from one reciprocal square root each
  fast_sqrt_rsqrt(a, l00, i00);
  const double l10 = b * i00, l20 = c * i00;
  const double t11 = d - l10 * l10;
  if (!(t11 > 0.0)) return false;
  fast_sqrt_rsqrt(t11, l11, i11);
  const double l21 = (e - l20 * l10) * i11;
  const double t22 = f - l20 * l20 - l21 * l21;
  if (!(t22 > 0.0)) return false;
  fast_sqrt_rsqrt(t22, l22, i22);
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  F[0] = i00; F[1] = i10; F[2] = i11; F[3] = i20; F[4] = i21; F[5] = i22;
  return true;
}

__host__ __device__ inline void sym3_mul(const double* S, double v0, double v1, double v2, double* o) {
  o[0] = S[0] * v0 + S[1] * v1 + S[2] * v2;
  o[1] = S[1] * v0 + S[3] * v1 + S[4] * v2;
  o[2] = S[2] * v0 + S[4] * v1 + S[5] * v2;
}

// packed index of the upper block triangle (i <= j) of an ncv x ncv block matrix
__host__ __device__ inline int64_t ut_block(int64_t i, int64_t j, int64_t ncv) {
  return i * ncv - (i * (i - 1)) / 2 + (j - i);
}
// Block-skyline storage of the upper block triangle of S: column j keeps the blocks (i, j), first[j] <= i <= j, at
// start[j] + (i - first[j]) — first[j] is the lowest camera slot that shares a landmark with slot j on ANY rank
// (ba_solver.hip build()), so every block a sweep can touch exists, and the buffer the ranks all-reduce holds the blocks
// that can be nonzero instead of all ncv (ncv + 1) / 2 (C4: 11 MB instead of 144 MB).
//
// Two forms.  index != NULL (up to kIndexMaxSlots slots): index[j * ns + i], i <= j, is the block's position or -1 — exactly
// the blocks of camera pairs that share a landmark on some rank, whatever the slot order (the nested-dissection order of
// chol_plan.h puts separator cameras last, whose skyline columns would be nearly full).  Otherwise the skyline proper:
// column j keeps the blocks first[j] <= i <= j.
struct BlockSky {
  const int32_t* first;
  const int64_t* start;
  const int32_t* index;
  int32_t ns;
};
constexpr int kIndexMaxSlots = 4096;
__host__ __device__ inline bool sky_has(const BlockSky& k, int64_t i, int64_t j) {
  return k.index ? k.index[j * k.ns + i] >= 0 : i >= k.first[j];
}
__host__ __device__ inline int64_t sky_block(const BlockSky& k, int64_t i, int64_t j) {
  return k.index ? (int64_t)k.index[j * k.ns + i] : k.start[j] + (i - k.first[j]);
}
// position of system column g in the slot-indexed vectors, -1 for a padding column or beyond the system
__host__ __device__ inline int vec_index(const int32_t* col_slot, int g, int n) {
  if (g >= n) return -1;
  if (!col_slot) return g;
  const int cs = col_slot[g];
  return cs < 0 ? -1 : (cs >> 3) * 6 + (cs & 7);
}
// packed index of lower-triangle tile (ti >= tj)
__host__ __device__ inline int64_t lt_tile(int64_t ti, int64_t tj) { return ti * (ti + 1) / 2 + tj; }

// ---- kernel argument blocks ---------------------------------------------------------------
struct SweepArgs {
  const ChunkHdr* chunks;
  const int32_t* chunk_cams;     // reduced-system slot of every local camera
  const int32_t* rec_cam;        // camera index (camera table row)
  const uint32_t* rec_meta;      // lcam | lpt<<8 | flags<<16
  const double* rec_xy;          // [nrec][2]
  const double* rec_d;           // log of the effective prior depth d*exp(s)+b
  const double* rec_m;           // depth loss magnitude
  const double* rec_a;           // depth loss scale
  const int32_t* pt_rec_start;   // [np+1]
  const uint16_t* pt_kv;         // [np] records with a variable camera (0: constant landmark)
  const uint32_t* blk_desc;      // per destination block: li | lj << 8
  const int32_t* blk_ent_start;  // [nblocks+1] entry range of each block
  const uint32_t* ents;          // per Schur pair: ri | rj << 8 | lpt << 16 (chunk-relative records)
  const double* camtab;          // [nc][24] at the linearisation point
  const double* pts;             // [np][3]
  const double* ps;              // [np][3] Jacobi scale (0: constant landmark)
  LossParams loss;
  double radius, min_diag, max_diag;
  int32_t ncv;
  int32_t dbg;  // timing-only ablation switches (0 in production)
  // long tracks
  const LongHdr* lhdr;
  int32_t nlong, nchunks;
  const int32_t* cam_slot;  // [nc] reduced-system slot or -1
  double* wl;               // W scratch of the long tracks
  double* slab;             // slabs of the dense chunks (k_track_sweep_dense -> k_reduce_slabs)
  int32_t chunk0;           // first chunk of this launch (blockIdx.x + chunk0 indexes `chunks` and `part`)
  int32_t pad0_;
  // outputs of the track sweep (accumulated, caller zeroes)
  BlockSky sky;    // where a block of S lives
  double* Sblk;    // block skyline of the upper block triangle, 36 doubles per block
  double* gc;      // [6 ncv] sum Jc^T r
  double* wv;      // [6 ncv] sum W Vinv g_p
  double* diagU;   // [6 ncv] diag(sum Jc^T Jc)
  double* part;    // [nchunks][4] cost, bad, max|g_p|, -
  double* diagV;   // DIAG mode: [np][3]
  // update sweep
  const double* yc;        // [6 ncv] reduced step (scaled coordinates)
  const double* camtab2;   // candidate cameras
  double* pts2;            // candidate landmarks
  double* part2;           // [nchunks][8]
  const LmCtl* ctl;        // inside a solve: trust-region radius and the stop flag live on the device (NULL: `radius` above)
  // k_track_sweep_dense as the first launch of an iteration (all chunks dense): what k_lm_prologue would do — an accepted
  // candidate becomes the state, the reduced buffer starts from zero — is spread over the sweep's workgroups
  int32_t adopt_on, adopt_nc;
  double* pts_rw; double* q_rw; double* t_rw; double* camtab_rw;
  const double* q2; const double* t2;
  double* red; int64_t nred;
};
// k_update_sweep with the camera update in front (all chunks dense): every workgroup forms the candidate rows of ITS chunk's
// cameras in LDS, workgroup 0 also does what k_cam_update does for all cameras (candidate poses and table, step and state norms,
// gradient maximum, the factorisation's failure flag) — one launch less on the iteration's critical path
struct CamUpdArgs {
  int32_t fuse, nc;
  const int32_t* cam_slot;     // [nc] slot or -1
  const int32_t* cam_of_slot;  // [ncv]
  const double* q; const double* t; const double* cs; const double* gc;
  const double* intr; const int32_t* intr_idx;
  double* q2; double* t2; double* camtab2; double* scal; int* chol_fail;
};
__device__ inline bool lm_accepted(const LmCtl* c) { return c != nullptr && __hip_atomic_load(&c->accepted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }

struct CostArgs {
  int64_t nrec;
  const int32_t* rec_cam; const int32_t* rec_pt; const uint32_t* rec_meta; const double* rec_xy;
  const double* rec_d; const double* rec_m; const double* rec_a;
  const double* camtab; const double* pts;
  LossParams loss;
  double* part;  // [gridDim.x][4]: reproj cost, depth cost, bad
};
// Caching device allocator (ba_solver.hip): hipMalloc / hipFree cost 10-300 us each and a handle makes ~60 of
// them; freed blocks are kept per device (up to a cap) and handed out again — with whatever an earlier owner
// left in them: every consumer initialises what it reads (MPSFM_POISON=1 fills each block with 0xFF to prove it).
void* cached_malloc(size_t bytes);
void cached_free(void* p);
// recycled non-blocking streams (ba_solver.hip); a released stream must be idle
hipError_t pooled_stream(hipStream_t* s);
void release_stream(hipStream_t s);

// second stream + events for the outer-panel look-ahead of the dense factorisation (dense_chol.hip)
struct DenseOverlap {
  int nb = 0;            // outer panel width in tile columns; 0: default (one panel up to 64 tile columns, else 8)
  bool big = true;       // LDS-staged 64x64 trailing update (false: per-tile workgroups, for A/B measurements)
  bool overlap = true;   // second-stream look-ahead
  bool no_level = false;    // per-step skyline path instead of the level-scheduled factorisation (A/B measurements)
  bool no_inverse = false;  // plain path: back substitution by groups instead of the inverse propagation (A/B measurements)
  hipStream_t s2 = nullptr;
  hipEvent_t evF[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t evB[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Device tables of the level-scheduled factorisation (chol_plan.h), static per handle.
struct LevelPlanDev {
  bool valid = false, use_pinv = false;
  const CholItem* d_items = nullptr;
  const int32_t* d_srcs = nullptr;
  const int32_t* d_rows = nullptr;
  const int32_t* d_struct_start = nullptr;
  const int32_t* d_struct_rows = nullptr;
  const int32_t* d_back_cols = nullptr;
  const int32_t* d_col_slot = nullptr;   // AssembleArgs::col_slot
  const int32_t* d_asm_tiles = nullptr;
  const uint8_t* d_tile_live = nullptr;  // [(nt+1)(nt+2)/2] 1 for the tiles in d_asm_tiles
  int32_t n_asm = 0, nlevels = 0;
  const int32_t* h_launch_start = nullptr;  // host: [nlevels + 1]
  const int32_t* h_back_start = nullptr;    // host: [nlevels + 1]
};

struct AssembleArgs {
  BlockSky sky;
  const double* Sblk; const double* gc; const double* wv; const double* diagU;
  int32_t ncv, n, nt;
  double radius, min_diag, max_diag;
  double* A;     // tiles
  double* Pinv;  // accumulators of the inverse propagation (same tile indexing as A), zeroed here; may be NULL
  const int32_t* tile_list;  // packed ids of the tiles to assemble (one workgroup each), NULL: all (nt+1)(nt+2)/2
  int32_t n_list;
  const int32_t* col_slot;   // [n] slot * 8 + coordinate of every column of the system, -1: padding column (identity row);
                             // NULL: column c belongs to slot c / 6 (no padding).  Vectors (gc, wv, diagU, y) are indexed by slot.
  int32_t n_vec;             // 6 x slots: length of those vectors
  double* y;                 // solution vector (slot-indexed), zeroed here when the back substitution accumulates into it (else NULL)
  const uint8_t* live;       // all tiles launched (inverse accumulators to zero): per packed id, 0 = the factorisation never
                             // reads this tile of A, skip the gather; NULL: every tile is assembled
  const LmCtl* ctl;  // as in SweepArgs
};


}  // namespace mpsfm
