// Host side of libmpsfm_hip: problem upload, chunking of the landmark tracks, the
// Levenberg-Marquardt control loop (Ceres 2.1 TrustRegionMinimizer semantics with the default
// pyceres.SolverOptions() that reference mpsfm/sfm/mapper/bundle_adjustment.py:285-293 uses) and
// the C ABI of include/mpsfm_hip.h.  All arithmetic on problem data runs in the HIP kernels of
// ba_kernels.hip / dense_chol.hip; this file only orders launches and takes the accept/reject
// decisions from a handful of scalars.
#include <algorithm>
#include <memory>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>

#include "common.h"
#include "devbuild.h"
#include "local_lm.h"

namespace mpsfm {

// ---- declarations of the launch wrappers (ba_kernels.hip, dense_chol.hip) ------------------------
void init_tile_tables(hipStream_t);
void launch_track_sweep(const SweepArgs&, int nchunks, bool diag_only, hipStream_t);
void launch_update_sweep(const SweepArgs&, int nchunks, hipStream_t, const CamUpdArgs* cu = nullptr);
void launch_track_sweep_dense(const SweepArgs&, int nchunks, hipStream_t);
void launch_reduce_slabs(const RedDest* dests, int ndest, const int32_t* srcs, const double* slab, double* Sblk, double* gc, double* wv, double* diagU,
                         const LmCtl* ctl, hipStream_t);
void launch_cost_records(const CostArgs&, int nblocks, hipStream_t);
void launch_reduce_cols(const double* part, int64_t rows, int stride, int ncols, uint32_t max_mask, double* out, hipStream_t,
                        double* out2 = nullptr, int gmax_slot = -1);
void launch_build_camtab(int nc, const double* q, const double* t, const double* intr, const int32_t* intr_idx,
                         const double* cs, double* camtab, hipStream_t);
void launch_cam_scales(int nc, const int32_t* cam_slot, const double* cmask, const double* diagU, int jacobi, double* cs, hipStream_t);
void launch_pt_scales(int64_t np, const uint16_t* pt_kv, const double* diagV, int jacobi, double* ps, hipStream_t);
void launch_cam_update(int nc, const int32_t* cam_slot, const double* q, const double* t, const double* cs, const double* yc,
                       const double* gc, double* q2, double* t2, double* scal, hipStream_t, const double* intr = nullptr,
                       const int32_t* intr_idx = nullptr, double* camtab2 = nullptr, int* chol_fail = nullptr, const LmCtl* ctl = nullptr);
void launch_lm_decide(LmCtl* ctl, double* scal, const LmOpts& o, LmCtl* host_copy, hipStream_t, const double* redsc = nullptr);
void launch_zero(double* p, int64_t n, const LmCtl* ctl, hipStream_t);
void launch_lm_reduce_decide(const double* part, const double* part2, int64_t rows, LmCtl* ctl, double* scal, const LmOpts& o, LmCtl* host_copy, hipStream_t);
void launch_lm_prologue(const LmCtl* ctl, double* red, int64_t nred, int nc, int64_t np, double* q, double* t, double* camtab, double* pts, const double* q2,
                        const double* t2, const double* camtab2, const double* pts2, hipStream_t);
void launch_lm_accept(const LmCtl* ctl, int nc, int64_t np, double* q, double* t, double* camtab, double* pts, const double* q2, const double* t2,
                      const double* camtab2, const double* pts2, hipStream_t);
void launch_pts_sqnorm(int64_t np, const uint16_t* pt_kv, const double* pts, double* part, int nblocks, hipStream_t);
void launch_gmax_to_slot(double* redsc, int rank, hipStream_t);
void launch_lm_pack(const double* scal, double* sums, hipStream_t);
void launch_lm_init(LmCtl* ctl, const double* scal, const double* sums, hipStream_t);
void launch_permute_pts(int64_t np, const int32_t* perm, const double* src, double* dst, bool scatter, hipStream_t);
void launch_gmax_from_slots(const double* redsc, double* scal, hipStream_t);
void launch_assemble(const AssembleArgs&, hipStream_t);
void launch_dense_solve(double* A, double* work, int nt, int n, double* y, int* fail, hipStream_t, DenseOverlap* ov, const LevelPlanDev* lp,
                        const LmCtl* ctl = nullptr);
bool dense_level(const DenseOverlap* ov, const LevelPlanDev* lp);
int dense_plain_max_tiles();
int dense_inv_rows();
size_t dense_work_doubles(int nt);
double* dense_pinv(double* work, int nt, const DenseOverlap* ov, const LevelPlanDev* lp);

thread_local std::string g_err;
extern int g_dbg_flags;  // dense_chol.hip: bits 0-7 dense-solve ablations, bits 8-15 track-sweep ablations
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(MPSFM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---- caching device allocator (declared in common.h) ---------------------------------------------------
namespace {
struct DevCache {
  static constexpr size_t kCap = (size_t)6 << 30;       // cached (free) bytes kept per device
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks[16];
  std::unordered_map<void*, std::pair<size_t, int>> live;  // pointer -> (block size, device)
  size_t cached[16] = {};
  static size_t block_size(size_t n) {  // 1/8-of-a-power-of-two granularity: sizes that differ a little share blocks
    n = std::max<size_t>(n, 256);
    size_t p = 256;
    while (p < n) p <<= 1;
    const size_t step = std::max<size_t>(p >> 3, 256);
    return (n + step - 1) / step * step;
  }
  void drop_all(int dev) {
    for (auto& kv : free_blocks[dev]) (void)hipFree(kv.second);
    free_blocks[dev].clear();
    cached[dev] = 0;
  }
  void* alloc(size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = std::min(std::max(dev, 0), 15);
    const size_t bs = block_size(bytes);
    std::lock_guard<std::mutex> lk(mu);
    auto it = free_blocks[dev].lower_bound(bs);
    if (it != free_blocks[dev].end() && it->first <= bs + bs / 4) {
      void* p = it->second;
      const size_t got = it->first;
      free_blocks[dev].erase(it);
      cached[dev] -= got;
      live[p] = {got, dev};
      poison(p, got);
      return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bs) != hipSuccess) {
      (void)hipGetLastError();
      drop_all(dev);  // give the cached blocks back and try once more
      if (hipMalloc(&p, bs) != hipSuccess) return nullptr;
    }
    live[p] = {bs, dev};
    poison(p, bs);
    return p;
  }
  // MPSFM_POISON=1 (tests): every block handed out is filled with 0xFF bytes (NaNs / huge indices), so a kernel
  // that reads memory nobody initialised fails loudly instead of finding the zeros a fresh hipMalloc often has
  static void poison(void* p, size_t n) {
    static const bool on = [] { const char* e = std::getenv("MPSFM_POISON"); return e && std::atoi(e) != 0; }();
    if (!on) return;
    (void)hipMemset(p, 0xFF, n);
    (void)hipDeviceSynchronize();
  }
  void release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(p);
    if (it == live.end()) { (void)hipFree(p); return; }
    const size_t bs = it->second.first;
    const int dev = it->second.second;
    live.erase(it);
    if (cached[dev] + bs > kCap) { (void)hipFree(p); return; }
    free_blocks[dev].emplace(bs, p);
    cached[dev] += bs;
  }
};
DevCache& dev_cache() {
  static DevCache* c = new DevCache();  // never destroyed: the HIP runtime may be gone at static-destruction time
  return *c;
}
}  // namespace
void* cached_malloc(size_t bytes) { return dev_cache().alloc(bytes); }
void cached_free(void* p) { dev_cache().release(p); }

// Streams, events and the small pinned scalar block of a handle are recycled the same way: creating and
// destroying them costs more than a whole solve of a small problem.  Per device; never destroyed.
namespace {
struct HandleResources {
  std::mutex mu;
  std::vector<hipStream_t> streams[16];
  std::vector<hipEvent_t> timing_events[16], plain_events[16];
  std::vector<void*> pinned[16];  // blocks of kPinnedBytes
  static constexpr size_t kPinnedBytes = 4096;
  static int dev() { int d = 0; (void)hipGetDevice(&d); return std::min(std::max(d, 0), 15); }
};
HandleResources& pool() { static HandleResources* r = new HandleResources(); return *r; }
}  // namespace
hipError_t pooled_stream(hipStream_t* s) {
  HandleResources& R = pool();
  { std::lock_guard<std::mutex> lk(R.mu); auto& v = R.streams[R.dev()]; if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; } }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
void release_stream(hipStream_t s) {
  if (!s) return;
  HandleResources& R = pool();
  std::lock_guard<std::mutex> lk(R.mu);
  R.streams[R.dev()].push_back(s);
}
static hipError_t pooled_event(hipEvent_t* e, bool timing) {
  HandleResources& R = pool();
  {
    std::lock_guard<std::mutex> lk(R.mu);
    auto& v = timing ? R.timing_events[R.dev()] : R.plain_events[R.dev()];
    if (!v.empty()) { *e = v.back(); v.pop_back(); return hipSuccess; }
  }
  return timing ? hipEventCreate(e) : hipEventCreateWithFlags(e, hipEventDisableTiming);
}
static void release_event(hipEvent_t e, bool timing) {
  if (!e) return;
  HandleResources& R = pool();
  std::lock_guard<std::mutex> lk(R.mu);
  (timing ? R.timing_events[R.dev()] : R.plain_events[R.dev()]).push_back(e);
}
static hipError_t pooled_pinned(void** p) {
  HandleResources& R = pool();
  { std::lock_guard<std::mutex> lk(R.mu); auto& v = R.pinned[R.dev()]; if (!v.empty()) { *p = v.back(); v.pop_back(); return hipSuccess; } }
  return hipHostMalloc(p, HandleResources::kPinnedBytes, hipHostMallocDefault);
}
static void release_pinned(void* p) {
  if (!p) return;
  HandleResources& R = pool();
  std::lock_guard<std::mutex> lk(R.mu);
  R.pinned[R.dev()].push_back(p);
}

// ---- RCCL, loaded at run time (no link-time dependency: single-GPU users never touch it) -----------------------------
// The four entry points the landmark-sharded solve needs.  dlopen finds the library already in the process (torch
// ships one) or the system's /opt/rocm copy.
namespace {
struct Rccl {
  typedef struct { char internal[128]; } UniqueId;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
  Rccl() {
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { why = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
    GetUniqueId = (int (*)(UniqueId*))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, UniqueId, int))dlsym(lib, "ncclCommInitRank");
    AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
    CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    ok = GetUniqueId && CommInitRank && AllReduce && CommDestroy;
    if (!ok) why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
  }
};
Rccl& rccl() { static Rccl* r = new Rccl(); return *r; }
constexpr int kNcclDouble = 8, kNcclSum = 0;  // ncclFloat64, ncclSum (rccl.h)
}  // namespace

template <typename T>
static int dev_alloc(T** p, size_t count) {
  if (count == 0) count = 1;
  *p = (T*)cached_malloc(count * sizeof(T));
  if (!*p) return fail(MPSFM_ENOMEM, "hipMalloc failed");
  return 0;
}
// Uploads of caller / table memory go through a process-wide pinned staging buffer (two halves, the host copy
// into one overlaps the DMA out of the other).  Handing pageable memory to hipMemcpy directly makes the
// runtime pin and later unpin every source range: measured 17 ms of stall after a 40 MB table upload.
// pageable -> pinned copy of one staging half: a single thread's memcpy (~20 GB/s here) is what bounded the uploads, not the bus;
// a few host threads in parallel (defined behind run_parts)
static void staged_copy(char* dst, const char* src, size_t n);
struct Stager {
  static constexpr size_t kHalf = (size_t)8 << 20;
  std::mutex mu;
  char* buf = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool busy[2] = {false, false};
  int next = 0;
  int init() {
    if (buf) return 0;  // set last: a partly created stager is torn down again and the next call retries
    char* b = nullptr;
    if (hipHostMalloc((void**)&b, 2 * kHalf, hipHostMallocDefault) != hipSuccess) return fail(MPSFM_ENOMEM, "hipHostMalloc (staging) failed");
    const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      for (auto& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
      if (st) (void)hipStreamDestroy(st);
      st = nullptr;
      (void)hipHostFree(b);
      return fail(MPSFM_EHIP, "creating the staging stream / events failed");
    }
    buf = b;
    return 0;
  }
  // blocking from the caller's point of view only at drain()
  int push(void* dst, const void* src, size_t bytes) {
    const char* s = (const char*)src;
    char* d = (char*)dst;
    while (bytes > 0) {
      const size_t n = std::min(bytes, kHalf);
      const int hf = next;
      next ^= 1;
      if (busy[hf]) { HIP_TRY(hipEventSynchronize(ev[hf])); busy[hf] = false; }
      staged_copy(buf + (size_t)hf * kHalf, s, n);
      HIP_TRY(hipMemcpyAsync(d, buf + (size_t)hf * kHalf, n, hipMemcpyHostToDevice, st));
      HIP_TRY(hipEventRecord(ev[hf], st));
      busy[hf] = true;
      s += n; d += n; bytes -= n;
    }
    return 0;
  }
  int drain() {
    HIP_TRY(hipStreamSynchronize(st));
    busy[0] = busy[1] = false;
    return 0;
  }
};
static Stager g_stagers[16];  // one per device ordinal
static Stager& stager() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_stagers[(size_t)std::min(std::max(dev, 0), 15)];
}

// for the other translation units (int_kernels.hip): queue a staged upload / wait for everything queued
int staged_upload(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  Stager& G = stager();
  std::lock_guard<std::mutex> lk(G.mu);
  if (int rc = G.init()) return rc;
  return G.push(dst, src, bytes);
}
int staged_drain() {
  Stager& G = stager();
  std::lock_guard<std::mutex> lk(G.mu);
  return G.buf ? G.drain() : 0;
}

// staged copy of host memory to the device, complete on return
static int staged_h2d(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  Stager& G = stager();
  std::lock_guard<std::mutex> lk(G.mu);
  if (int rc = G.init()) return rc;
  if (int rc = G.push(dst, src, bytes)) return rc;
  return G.drain();
}

template <typename T>
static int dev_upload(T** p, const std::vector<T>& v) {
  int rc = dev_alloc(p, v.size());
  if (rc) return rc;
  if (v.empty()) return 0;
  Stager& G = stager();
  std::lock_guard<std::mutex> lk(G.mu);
  if ((rc = G.init())) return rc;
  return G.push(*p, v.data(), v.size() * sizeof(T));  // build() drains once after the last table
}
// Large host blocks of the table build come from a process-wide cache: a fresh 40 MB block costs its page faults on first
// touch and an munmap on release (several ms per create at C3); a recycled one costs neither.  Power-of-two buckets from
// 1 MB, at most 512 MB kept (MPSFM_HOST_CACHE_MB).
struct HostBlockCache {
  static constexpr size_t kMinBytes = size_t(1) << 20;
  const size_t kHostCacheBytes = [] {  // MPSFM_HOST_CACHE_MB: how much released host memory is kept for the next build (0: none)
    const char* e = std::getenv("MPSFM_HOST_CACHE_MB");
    return (size_t)((e && std::atoi(e) >= 0) ? std::atoi(e) : 512) << 20;
  }();
  std::mutex mu;
  std::vector<std::pair<size_t, void*>> free_blocks;
  size_t cached = 0;
  static size_t bucket(size_t bytes) { size_t b = kMinBytes; while (b < bytes) b <<= 1; return b; }
  void* take(size_t bytes, size_t& got) {
    if (bytes < kMinBytes) { got = 0; return ::operator new(std::max<size_t>(bytes, 1)); }
    got = bucket(bytes);
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < free_blocks.size(); ++i)
        if (free_blocks[i].first == got) {
          void* p = free_blocks[i].second;
          free_blocks[i] = free_blocks.back(); free_blocks.pop_back();
          cached -= got;
          return p;
        }
    }
    return ::operator new(got);
  }
  void give(void* p, size_t got) {
    if (!p) return;
    if (got) {
      std::lock_guard<std::mutex> lk(mu);
      if (cached + got <= kHostCacheBytes) { free_blocks.emplace_back(got, p); cached += got; return; }
    }
    ::operator delete(p);
  }
  ~HostBlockCache() { for (auto& b : free_blocks) ::operator delete(b.second); }
};
static HostBlockCache& host_cache() { static HostBlockCache c; return c; }

// uninitialised host array of trivially copyable elements (std::vector would zero-fill tens of MB on one thread)
template <typename T>
struct HostBuf {
  static_assert(std::is_trivially_copyable<T>::value && std::is_trivially_destructible<T>::value, "HostBuf holds raw storage");
  T* p = nullptr;
  size_t n = 0, got = 0;
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { host_cache().give(p, got); }
  void alloc(size_t k) {
    host_cache().give(p, got);
    p = static_cast<T*>(host_cache().take(std::max<size_t>(k, 1) * sizeof(T), got));
    n = k;
  }
  void release() { host_cache().give(p, got); p = nullptr; n = got = 0; }
  size_t size() const { return n; }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};
template <typename T>
static int dev_upload(T** p, HostBuf<T>& v) {
  int rc = dev_alloc(p, v.size());
  if (rc) return rc;
  if (v.size() == 0) return 0;
  Stager& G = stager();
  std::lock_guard<std::mutex> lk(G.mu);
  if ((rc = G.init())) return rc;
  return G.push(*p, v.data(), v.size() * sizeof(T));
}
static int drain_uploads() {
  Stager& G = stager();
  std::lock_guard<std::mutex> lk(G.mu);
  return G.buf ? G.drain() : 0;
}

}  // namespace mpsfm

using namespace mpsfm;

struct mpsfm_ba_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  void* comm = nullptr;  // ncclComm_t of a landmark-sharded run with use_rccl
  DenseOverlap ov;  // second stream for the dense factorisation in outer panels (MPSFM_CHOL_NB)
  std::vector<int32_t> sky_first;   // block skyline of S (BlockSky), host copies
  std::vector<int64_t> sky_start;
  int32_t* d_sky_first = nullptr;
  int64_t* d_sky_start = nullptr;
  std::vector<int32_t> sky_index;   // index form of BlockSky (up to kIndexMaxSlots slots), host copy
  int32_t* d_sky_index = nullptr;
  CholPlan plan;                    // camera order, tile elimination tree and launch tables of the dense factorisation (chol_plan.h)
  LevelPlanDev lp;
  CholItem* d_lp_items = nullptr;
  uint8_t* d_lp_live = nullptr;
  int32_t* d_lp_col_slot = nullptr;
  int32_t *d_lp_srcs = nullptr, *d_lp_rows = nullptr, *d_lp_struct_start = nullptr, *d_lp_struct_rows = nullptr, *d_lp_back_cols = nullptr, *d_lp_asm = nullptr;
  std::vector<int32_t> nat_slot;    // variable camera in the caller's order -> slot (the accessors of S and y speak the caller's order)
  int n_user = 0;                   // 6 x variable cameras: the reduced dimension the caller sees and the length of the slot-indexed vectors (n counts the
                                    // system's columns incl. the alignment padding)
  bool own_stream = false;
  mpsfm_ba_options opt{};
  LossParams loss{};
  // sizes
  int nc = 0, np_user = 0;
  int64_t np = 0, np_chunked = 0;   // re-ordered landmarks (all referenced) / those inside chunks
  int64_t nrec = 0, nfixed = 0, nblocks_total = 0, nblocks_reduced = 0;
  double nblocks_global = 0, nblocks_reduced_global = 0, nvarpts_global = 0;
  int ncv = 0, n = 0, nt = 0, nchunks = 0, nlong = 0;
  int n_dense = 0;                  // chunks [0, n_dense) are swept by k_track_sweep_dense, the rest by the general kernel
  double* d_slab = nullptr;         // slabs of the dense chunks
  RedDest* d_red_dests = nullptr;   // slab reduction: destination parts and their sources
  int32_t* d_red_srcs = nullptr;
  int n_red_dests = 0;
  int64_t n_red_srcs = 0, n_chunk_cams = 0, n_blk_desc = 0, n_blk_ent_start = 0, n_ents = 0;  // table sizes (diagnostics: mpsfm_debug_table)
  bool built_on_device = false;
  LongHdr* d_lhdr = nullptr;
  double* d_wl = nullptr;
  int64_t red_count = 0, sblk_count = 0, sblk_blocks = 0;
  std::vector<int32_t> perm;        // re-ordered landmark -> caller's index
  int32_t* d_cam_of_slot = nullptr; // slot -> camera (the fused camera update of k_update_sweep)
  int32_t* d_perm = nullptr;        // device copy, and the landmarks in the caller's order as last uploaded: the state crosses the bus
  double* d_user_pts = nullptr;     // unpermuted and is re-ordered on the device (every landmark referenced: np == np_user)
  std::vector<int32_t> cam_slot_h;
  // device state
  double *d_q = nullptr, *d_t = nullptr, *d_q2 = nullptr, *d_t2 = nullptr, *d_q0 = nullptr, *d_t0 = nullptr;
  double *d_pts = nullptr, *d_pts2 = nullptr, *d_pts0 = nullptr;
  double *d_intr = nullptr, *d_cmask = nullptr, *d_cs = nullptr, *d_camtab = nullptr, *d_camtab2 = nullptr;
  int32_t *d_intr_idx = nullptr, *d_cam_slot = nullptr;
  double *d_ps = nullptr, *d_diagV = nullptr;
  ChunkHdr* d_chunks = nullptr;
  int32_t *d_chunk_cams = nullptr, *d_rec_cam = nullptr, *d_rec_pt = nullptr, *d_pt_rec_start = nullptr, *d_blk_ent_start = nullptr;
  uint32_t *d_blk_desc = nullptr, *d_ents = nullptr;
  uint32_t* d_rec_meta = nullptr;
  uint16_t* d_pt_kv = nullptr;
  double *d_rec_xy = nullptr, *d_rec_d = nullptr, *d_rec_m = nullptr, *d_rec_a = nullptr;
  // fixed blocks (constant camera and constant landmark)
  int32_t *d_fx_cam = nullptr, *d_fx_pt = nullptr;
  uint32_t* d_fx_meta = nullptr;
  double *d_fx_xy = nullptr, *d_fx_d = nullptr, *d_fx_m = nullptr, *d_fx_a = nullptr;
  // reduced buffer: Sblk | gc | wv | diagU | scalars
  double* d_red = nullptr;
  double *d_Sblk = nullptr, *d_gc = nullptr, *d_wv = nullptr, *d_diagU = nullptr, *d_redsc = nullptr;
  double *d_part = nullptr, *d_part2 = nullptr, *d_scal = nullptr, *d_costpart = nullptr;
  double* h_scal = nullptr;  // pinned
  LmCtl* d_ctl = nullptr;    // Levenberg-Marquardt control block (device) and the two pinned slots its copies land in
  LmCtl* h_ctl = nullptr;
  hipEvent_t ev2[4] = {nullptr, nullptr, nullptr, nullptr};  // second set of phase events (two iterations are in flight)
  double *d_A = nullptr, *d_yc = nullptr, *d_dwork = nullptr;
  int* d_fail = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  double last_radius = 1e4;
  bool scales_ready = false;
  // single-launch solver of small problems (local_lm.hip): two accumulators | barrier words + clocks | per-iteration heads
  bool local_ok = false;
  double* d_local_acc = nullptr;
  int64_t* d_local_sync = nullptr;   // [0]: two 32-bit barrier words, [1..4]: phase clocks
  LmHead* d_local_log = nullptr;
  int local_log_cap = 0;
};

namespace mpsfm {

static void free_handle(mpsfm_ba_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  // Error returns of the solve (a failing all-reduce hook, a HIP error) and reset_state + destroy leave copies and
  // kernels in flight: both streams must be idle before the blocks go back to the process-wide cache, where a
  // handle on another stream or host thread may receive them at once.
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->ov.s2) (void)hipStreamSynchronize(h->ov.s2);
  void* ptrs[] = {h->d_q, h->d_t, h->d_q2, h->d_t2, h->d_q0, h->d_t0, h->d_pts, h->d_pts2, h->d_pts0, h->d_intr, h->d_cmask,
                  h->d_cs, h->d_camtab, h->d_camtab2, h->d_intr_idx, h->d_cam_slot, h->d_ps, h->d_diagV, h->d_chunks,
                  h->d_chunk_cams, h->d_rec_cam, h->d_rec_pt, h->d_pt_rec_start, h->d_blk_ent_start, h->d_blk_desc, h->d_ents, h->d_rec_meta, h->d_pt_kv,
                  h->d_rec_xy, h->d_rec_d, h->d_rec_m, h->d_rec_a, h->d_fx_cam, h->d_fx_pt, h->d_fx_meta, h->d_fx_xy, h->d_fx_d,
                  h->d_fx_m, h->d_fx_a, h->d_red, h->d_part, h->d_part2, h->d_scal, h->d_costpart, h->d_A, h->d_yc, h->d_dwork, h->d_fail, h->d_lhdr, h->d_wl, h->d_slab, h->d_red_dests, h->d_red_srcs,
                  h->d_sky_first, h->d_sky_start, h->d_sky_index,
                  h->d_local_acc, h->d_local_sync, h->d_local_log, h->d_perm, h->d_user_pts, h->d_cam_of_slot,
                  h->d_lp_items, h->d_lp_srcs, h->d_lp_rows, h->d_lp_struct_start, h->d_lp_struct_rows, h->d_lp_back_cols, h->d_lp_asm, h->d_lp_live, h->d_lp_col_slot};
  for (void* p : ptrs) cached_free(p);
  if (h->comm) (void)rccl().CommDestroy(h->comm);
  release_pinned(h->h_scal);
  release_pinned(h->h_ctl);
  cached_free(h->d_ctl);
  for (auto& e : h->ev2) release_event(e, true);
  for (auto& e : h->ev) release_event(e, true);
  for (auto& e : h->ov.evF) release_event(e, false);
  for (auto& e : h->ov.evB) release_event(e, false);
  release_stream(h->ov.s2);
  if (h->own_stream) release_stream(h->stream);
  delete h;
}

static int check_problem(const mpsfm_ba_problem* P) {
  if (!P) return fail(MPSFM_EINVAL, "problem is NULL");
  if (P->n_cams < 0 || P->n_pts < 0 || P->n_intr < 0 || P->n_obs < 0 || P->n_dobs < 0) return fail(MPSFM_EINVAL, "negative size");
  if (P->n_cams > 0 && (!P->cam_intr_idx || !P->pose_const || !P->cam_intr)) return fail(MPSFM_EINVAL, "camera arrays are NULL");
  if (P->n_pts > 0 && !P->pt_const) return fail(MPSFM_EINVAL, "pt_const is NULL");
  if (P->n_obs > 0 && (!P->obs_cam || !P->obs_pt || !P->obs_xy)) return fail(MPSFM_EINVAL, "observation arrays are NULL");
  if (P->n_dobs > 0 && (!P->dobs_cam || !P->dobs_pt || !P->dobs_depth || !P->dobs_magnitude || !P->dobs_param))
    return fail(MPSFM_EINVAL, "depth observation arrays are NULL");
  if (P->gauge_axis_cam < -1 || P->gauge_axis_cam >= P->n_cams) return fail(MPSFM_EINVAL, "gauge_axis_cam out of range");
  for (int i = 0; i < P->n_cams; ++i)
    if (P->cam_intr_idx[i] < 0 || P->cam_intr_idx[i] >= P->n_intr) return fail(MPSFM_EINVAL, "cam_intr_idx out of range");
  // branch-free sweeps (they vectorise; 7.6 M blocks at C4): an index is in range when it is below the bound as an unsigned number
  auto out_of_range = [](const int32_t* v, int64_t n, int32_t bound) {
    uint32_t bad = 0;
    const uint32_t b = (uint32_t)bound;
    for (int64_t i = 0; i < n; ++i) bad |= (uint32_t)((uint32_t)v[i] >= b);
    return bad != 0;
  };
  if (out_of_range(P->obs_cam, P->n_obs, P->n_cams) || out_of_range(P->obs_pt, P->n_obs, P->n_pts))
    return fail(MPSFM_EINVAL, "observation index out of range");
  if (out_of_range(P->dobs_cam, P->n_dobs, P->n_cams) || out_of_range(P->dobs_pt, P->n_dobs, P->n_pts))
    return fail(MPSFM_EINVAL, "depth observation index out of range");
  for (int t : {P->reproj_loss_type, P->depth_loss_type})
    if (t < MPSFM_LOSS_TRIVIAL || t > MPSFM_LOSS_CAUCHY) return fail(MPSFM_EINVAL, "unknown loss type");
  return 0;
}

static bool sharded(const mpsfm_ba_handle* h) { return h->opt.allreduce != nullptr || h->comm != nullptr; }
static int rccl_allreduce(mpsfm_ba_handle* h, double* dbuf, int64_t count) {
  const int rc = rccl().AllReduce(dbuf, dbuf, (size_t)count, kNcclDouble, kNcclSum, h->comm, h->stream);
  if (rc != 0) return fail(MPSFM_ECOMM, std::string("ncclAllReduce: ") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "failed"));
  return 0;
}
static int allreduce_host(mpsfm_ba_handle* h, double* buf, int64_t count) {
  if (count <= 0) return 0;
  if (h->comm) {  // host values travel through a device scratch block
    double* d = (double*)cached_malloc(sizeof(double) * (size_t)count);
    if (!d) return fail(MPSFM_ENOMEM, "hipMalloc failed");
    int rc = 0;
    if (hipMemcpyAsync(d, buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail(MPSFM_EHIP, "hipMemcpyAsync failed");
    if (!rc) rc = rccl_allreduce(h, d, count);
    if (!rc && hipMemcpyAsync(buf, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = fail(MPSFM_EHIP, "hipMemcpyAsync failed");
    (void)hipStreamSynchronize(h->stream);
    cached_free(d);
    return rc;
  }
  if (!h->opt.allreduce) return 0;
  if (h->opt.allreduce(h->opt.allreduce_user, buf, count, 0, nullptr)) return fail(MPSFM_ECOMM, "all-reduce hook failed (host buffer)");
  return 0;
}
static int allreduce_dev(mpsfm_ba_handle* h, double* buf, int64_t count) {
  if (count <= 0) return 0;
  if (h->comm) return rccl_allreduce(h, buf, count);
  if (!h->opt.allreduce) return 0;
  if (h->opt.allreduce(h->opt.allreduce_user, buf, count, 1, (void*)h->stream)) return fail(MPSFM_ECOMM, "all-reduce hook failed (device buffer)");
  return 0;
}

struct Blk { int32_t cam; int32_t key; uint8_t kind; int64_t src; };

// ---- host threads for the table build (plain std::thread: no OpenMP runtime beside torch's) -------
// CPUs this process may use: scheduler affinity capped by the cgroup quota; MPSFM_HOST_THREADS overrides.
static int host_threads() {
  static const int n = [] {
    if (const char* e = std::getenv("MPSFM_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) return std::min(v, 64); }
    int cpus = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::max(1, CPU_COUNT(&set));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char a[64]; double per = 0.0;
      if (std::fscanf(f, "%63s %lf", a, &per) == 2 && std::strcmp(a, "max") != 0 && per > 0.0)
        cpus = std::min(cpus, std::max(1, (int)(std::atof(a) / per + 0.5)));
      std::fclose(f);
    }
    return std::min(cpus, 32);
  }();
  return n;
}
// Persistent workers for the phases of the table build: creating and joining 15 threads costs ~0.4 ms, and one create runs
// a dozen phases.  One job at a time; a caller that finds the pool busy (another handle being created) starts plain threads
// as before.  Parts are claimed with the job's generation, so a worker that wakes late never touches a newer job; a forked
// child starts over with a pool of its own (the parent's workers do not exist there).
class HostPool {
 public:
  typedef void (*Call)(void* ctx, int part, int nparts);
  // false: the pool is busy, nothing ran
  bool try_run(int nparts, Call call, void* ctx) {
    std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
    if (!job.owns_lock()) return false;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (workers_.empty()) {
        const int nw = std::max(host_threads() - 1, 1);
        for (int i = 0; i < nw; ++i) workers_.emplace_back([this] { work(); });
      }
      call_ = call; ctx_ = ctx; nparts_ = nparts;
      done_.store(0, std::memory_order_relaxed);
      ++gen_;
      state_.store((gen_ << 32) | 1u, std::memory_order_release);  // part 0 is the caller's
    }
    cv_work_.notify_all();
    call(ctx, 0, nparts);
    finish_part();
    claim_loop(gen_, call, ctx, nparts);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return done_.load(std::memory_order_acquire) == nparts; });
    return true;
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_work_.notify_all();
    for (auto& w : workers_) w.join();
  }

 private:
  void finish_part() {
    if (done_.fetch_add(1, std::memory_order_acq_rel) + 1 == nparts_) { std::lock_guard<std::mutex> lk(mu_); cv_done_.notify_all(); }
  }
  void claim_loop(uint64_t gen, Call call, void* ctx, int nparts) {
    uint64_t s = state_.load(std::memory_order_acquire);
    while ((s >> 32) == gen && (int)(s & 0xffffffffu) < nparts) {
      if (state_.compare_exchange_weak(s, s + 1, std::memory_order_acq_rel)) {
        call(ctx, (int)(s & 0xffffffffu), nparts);
        finish_part();
        s = state_.load(std::memory_order_acquire);
      }
    }
  }
  void work() {
    uint64_t seen = 0;
    for (;;) {
      Call call; void* ctx; int nparts; uint64_t gen;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen = gen_; call = call_; ctx = ctx_; nparts = nparts_;
      }
      claim_loop(gen, call, ctx, nparts);
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::thread> workers_;
  std::atomic<uint64_t> state_{0};
  std::atomic<int> done_{0};
  uint64_t gen_ = 0;
  Call call_ = nullptr; void* ctx_ = nullptr; int nparts_ = 0;
  bool stop_ = false;
};
static std::atomic<HostPool*> g_host_pool{nullptr};
static HostPool* host_pool() {
  static std::once_flag once;
  std::call_once(once, [] {
    pthread_atfork(nullptr, nullptr, [] { g_host_pool.store(nullptr); });  // child: the old pool is abandoned, never destroyed
    std::atexit([] { delete g_host_pool.exchange(nullptr); });
  });
  HostPool* p = g_host_pool.load(std::memory_order_acquire);
  if (!p) {
    HostPool* fresh = new HostPool();
    if (g_host_pool.compare_exchange_strong(p, fresh)) p = fresh; else delete fresh;
  }
  return p;
}
// f(part, nparts) on nparts threads (the calling thread takes part 0)
template <class F>
static void run_parts(int nparts, F&& f) {
  if (nparts <= 1) { f(0, std::max(nparts, 1)); return; }
  static const bool use_pool = !(std::getenv("MPSFM_HOST_POOL") && std::atoi(std::getenv("MPSFM_HOST_POOL")) == 0);
  typedef typename std::remove_reference<F>::type Fn;
  if (use_pool && host_pool()->try_run(nparts, [](void* c, int t, int n) { (*static_cast<Fn*>(c))(t, n); }, const_cast<void*>(static_cast<const void*>(&f)))) return;
  std::vector<std::thread> th;
  th.reserve((size_t)std::max(nparts - 1, 0));
  for (int t = 1; t < nparts; ++t) th.emplace_back([&f, t, nparts] { f(t, nparts); });
  f(0, nparts);
  for (auto& x : th) x.join();
}
// f(begin, end) over [0, n) cut into nearly equal contiguous parts
template <class F>
static void parallel_ranges(int64_t n, int64_t min_grain, F&& f) {
  const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / std::max<int64_t>(min_grain, 1)));
  if (parts <= 1) { f((int64_t)0, n); return; }
  run_parts(parts, [&](int t, int np) { f(n * t / np, n * (t + 1) / np); });
}

static void staged_copy(char* dst, const char* src, size_t n) {
  constexpr size_t kGrain = (size_t)1 << 20;
  static const int max_threads = [] { const char* e = std::getenv("MPSFM_STAGE_THREADS"); return e ? std::max(std::atoi(e), 1) : 6; }();
  const int parts = (int)std::min<size_t>((size_t)std::min(host_threads(), max_threads), n / kGrain);
  if (parts <= 1) { std::memcpy(dst, src, n); return; }
  run_parts(parts, [&](int t, int np) {
    const size_t a = (n * (size_t)t / (size_t)np) & ~(size_t)63, b = t + 1 == np ? n : ((n * (size_t)(t + 1) / (size_t)np) & ~(size_t)63);
    std::memcpy(dst + a, src + a, b - a);
  });
}

// Build the re-ordered, chunked record tables and upload everything.
// the tables of the level-scheduled factorisation (h->plan) to the device
// The camera graph of a landmark-sharded run is the UNION over the ranks, and the exchange can only SUM doubles: every rank
// packs its adjacency bits as indicator digits in base (world + 1), E digits per double (E chosen so that a sum of `world`
// such numbers stays below 2^53, i.e. exact), the packed vectors are summed, and a digit > 0 means "some rank has the edge".
static int graph_digits(int world) {
  int E = 1;
  double cap = 9007199254740992.0 / (world + 1);
  while (cap >= (world + 1) && E < 16) { cap /= (world + 1); ++E; }
  return E;
}
static void pack_graph(const CamGraph& graph, int world, std::vector<double>& packed) {
  const int E = graph_digits(world), n = graph.n;
  const int64_t nbits = (int64_t)n * n;
  packed.assign((size_t)((nbits + E - 1) / E), 0.0);
  double pw[16];
  pw[0] = 1.0;
  for (int e = 1; e < 16; ++e) pw[e] = pw[e - 1] * (double)(world + 1);
  for (int a = 0; a < n; ++a) {
    const uint64_t* row = graph.row(a);
    for (int w = 0; w < graph.words; ++w) {
      uint64_t m = row[w];
      while (m) {
        const int64_t q = (int64_t)a * n + (w * 64 + __builtin_ctzll(m));
        m &= m - 1;
        packed[(size_t)(q / E)] += pw[q % E];
      }
    }
  }
}
static void unpack_graph(const std::vector<double>& packed, int world, CamGraph& graph) {
  const int E = graph_digits(world), n = graph.n;
  const int64_t nbits = (int64_t)n * n;
  for (size_t w = 0; w < packed.size(); ++w) {
    double v = packed[w];
    for (int e = 0; e < E && v > 0.0; ++e) {
      const double d = std::fmod(v, (double)(world + 1));
      v = std::floor(v / (world + 1));
      const int64_t q = (int64_t)w * E + e;
      if (d > 0.0 && q < nbits) graph.set((int)(q / n), (int)(q % n));
    }
  }
}

static int upload_plan(mpsfm_ba_handle* h, int64_t nblk) {
  const CholPlan& PL = h->plan;
  int rc2 = 0;
  if ((rc2 = dev_upload(&h->d_lp_items, PL.items))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_srcs, PL.srcs))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_rows, PL.rows))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_struct_start, PL.struct_start))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_struct_rows, PL.struct_rows))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_back_cols, PL.back_cols))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_asm, PL.asm_tiles))) return rc2;
  if ((rc2 = dev_upload(&h->d_lp_col_slot, PL.slot_of_col))) return rc2;
  {
    std::vector<uint8_t> live((size_t)(PL.nt + 1) * (size_t)(PL.nt + 2) / 2, 0);
    for (int32_t id : PL.asm_tiles) live[(size_t)id] = 1;
    if ((rc2 = dev_upload(&h->d_lp_live, live))) return rc2;
  }
  LevelPlanDev& D = h->lp;
  D.valid = PL.nt >= 1 && PL.nlevels >= 1; D.use_pinv = PL.use_pinv;
  D.d_items = h->d_lp_items; D.d_srcs = h->d_lp_srcs; D.d_rows = h->d_lp_rows;
  D.d_struct_start = h->d_lp_struct_start; D.d_struct_rows = h->d_lp_struct_rows; D.d_back_cols = h->d_lp_back_cols;
  D.d_asm_tiles = h->d_lp_asm; D.d_tile_live = h->d_lp_live; D.d_col_slot = PL.slot_of_col.empty() ? nullptr : h->d_lp_col_slot; D.n_asm = (int32_t)PL.asm_tiles.size(); D.nlevels = PL.nlevels;
  D.h_launch_start = PL.launch_start.data(); D.h_back_start = PL.back_start.data();
  if (h->opt.verbose >= 2)
    std::fprintf(stderr, "[mpsfm_ba] build: camera order: %s (depth %d), %d slots for %d cameras, %d tile columns in %d levels, %lld tile products, %lld inverse roles, %d blocks of S\n",
                 PL.nd_depth < 0 ? "caller's" : "nested dissection", PL.nd_depth, PL.nslots, PL.ncv, PL.nt, PL.nlevels, (long long)PL.products, (long long)PL.roles, (int)nblk);
  return 0;
}

// Pair tables of ONE chunk for the general kernel, appended to (blk_desc, ents, blk_ent_start): the Schur pairs of its variable
// landmarks grouped by the 6x6 destination block (counting sort, heaviest blocks first), cut into work items of at most kItemPairs
// pairs; a dense chunk only gets its sentinel.  `rec_meta` is indexed by the global record, `pt_kv` / `pt_rec_start` / `order` by the
// re-ordered landmark.  Shared by the host build and by the device build's general chunks.
struct PairEnt { uint16_t key; uint32_t ent; };
struct PairScratch {
  std::vector<PairEnt> pe, pe_sorted;
  std::vector<std::pair<int, int>> blk_order, items;  // (count, first index into pe)
  std::vector<int32_t> cnt;
};
static void append_pair_tables(ChunkHdr& H, const uint32_t* rec_meta, const uint16_t* pt_kv, const int32_t* pt_rec_start, const int32_t* order,
                               const uint8_t* pt_const, std::vector<uint32_t>& o_blk_desc, std::vector<uint32_t>& o_ents, std::vector<int32_t>& o_blk_ent_start,
                               PairScratch& S) {
  std::vector<PairEnt>&pe = S.pe, &pe_sorted = S.pe_sorted;
  std::vector<std::pair<int, int>>&blk_order = S.blk_order, &items = S.items;
  std::vector<int32_t>& cnt = S.cnt;
  const int64_t c_first = H.pt0, end_pt = (int64_t)H.pt0 + H.npt;
  pe.clear();
  if (!H.dense) {
    for (int64_t k = c_first; k < end_pt; ++k) {
      const int p = order[k];
      if (pt_const[p]) continue;
      // Schur pairs of this landmark: records rbase .. rbase+kv-1 have variable cameras (slot-sorted)
      const int rbase = pt_rec_start[(size_t)k] - H.rec0;
      const int kv = (int)pt_kv[(size_t)k];
      const uint32_t lpt = (uint32_t)(k - c_first);
      for (int i = 0; i < kv; ++i) {
        const uint32_t li = rec_meta[(size_t)H.rec0 + rbase + i] & 0xff;
        for (int j = i; j < kv; ++j) {
          const uint32_t lj = rec_meta[(size_t)H.rec0 + rbase + j] & 0xff;
          pe.push_back(PairEnt{(uint16_t)(li | (lj << 8)), (uint32_t)(rbase + i) | ((uint32_t)(rbase + j) << 8) | (lpt << 16)});
          // two records of one camera: the diagonal block needs B + B^T
          if (li == lj && i != j)
            pe.push_back(PairEnt{(uint16_t)(li | (lj << 8)), (uint32_t)(rbase + j) | ((uint32_t)(rbase + i) << 8) | (lpt << 16)});
        }
      }
    }
  }
  // group the pairs by destination block (counting sort on li*ncam+lj); heaviest blocks first
  {
    const int nl = std::max(H.ncam, 1);
    cnt.assign((size_t)nl * nl + 1, 0);
    for (const PairEnt& e : pe) cnt[(size_t)(e.key & 0xff) * nl + (e.key >> 8) + 1]++;
    for (size_t q = 1; q < cnt.size(); ++q) cnt[q] += cnt[q - 1];
    pe_sorted.resize(pe.size());
    for (const PairEnt& e : pe) pe_sorted[(size_t)cnt[(size_t)(e.key & 0xff) * nl + (e.key >> 8)]++] = e;
    pe.swap(pe_sorted);
  }
  blk_order.clear();
  for (size_t i = 0; i < pe.size();) {
    size_t j = i;
    while (j < pe.size() && pe[j].key == pe[i].key) ++j;
    blk_order.emplace_back((int)(j - i), (int)i);
    i = j;
  }
  std::stable_sort(blk_order.begin(), blk_order.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first > y.first; });
  // work items: runs of at most kItemPairs pairs of one block.  Block-major: all items of a block are
  // neighbours, so the flush combines them (one atomic pass per block and round)
  items.clear();
  for (const auto& bo : blk_order)
    for (int q = 0; q < bo.first; q += kItemPairs) items.emplace_back(std::min(kItemPairs, bo.first - q), bo.second + q);
  H.blk0 = (int32_t)o_blk_desc.size();  // thread-local for now
  H.ent0 = (int32_t)o_ents.size();
  H.nent = (int32_t)pe.size();
  H.nblk = (int32_t)items.size();
  for (const auto& it : items) {
    o_blk_desc.push_back(pe[(size_t)it.second].key);
    o_blk_ent_start.push_back((int32_t)(o_ents.size() - (size_t)H.ent0));
    for (int q = 0; q < it.first; ++q) o_ents.push_back(pe[(size_t)(it.second + q)].ent);
  }
  o_blk_ent_start.push_back((int32_t)(o_ents.size() - (size_t)H.ent0));  // per-chunk sentinel
}

static int build(mpsfm_ba_handle* h, const mpsfm_ba_problem* P, const mpsfm_ba_state* st) {
  const int nc = P->n_cams, npu = P->n_pts;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (h->opt.verbose < 2) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mpsfm_ba] build: %-28s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  h->nc = nc; h->np_user = npu;
  h->loss.reproj_type = P->reproj_loss_type; h->loss.reproj_a = P->reproj_loss_scale;
  h->loss.reproj_mag = P->reproj_loss_magnitude; h->loss.depth_type = P->depth_loss_type;

  // -- Device-side table build (build_dev.hip) where it applies: one rank, at most kIndexMaxSlots non-constant cameras, no
  //    landmark with more blocks than a chunk holds, no chunk for the general kernel.  Stage 1 runs here (block counts per camera,
  //    blocks grouped by landmark, camera graph); the rest of this function then skips its host phases.  MPSFM_DEV_BUILD=0: host.
  bool dev = false;
  std::unique_ptr<DevBuilder> devb;
  DevBuildOut DB;
  std::vector<uint64_t> dev_gbits;
  std::vector<int32_t> prov((size_t)std::max(nc, 1), -1);  // provisional slots of the graph stage: the non-constant cameras in order
  int nprov = 0;
  for (int i = 0; i < nc; ++i) if (!P->pose_const[i]) prov[(size_t)i] = nprov++;
  const int dev_words = (nprov + 63) / 64;
  // -- cameras of the reduced program: not constant and referenced by a residual block (any shard)
  std::vector<double> cnt(nc + 1, 0.0);
  const bool dev_wanted = !sharded(h) && nprov <= kIndexMaxSlots && nc <= 8192 && P->n_obs + P->n_dobs > 0 &&
                          !(std::getenv("MPSFM_DEV_BUILD") && std::atoi(std::getenv("MPSFM_DEV_BUILD")) == 0) &&
                          !(std::getenv("MPSFM_CHOL_GRAPH") && std::atoi(std::getenv("MPSFM_CHOL_GRAPH")) == 0);
  if (dev_wanted) {
    devb.reset(new DevBuilder());
    int64_t max_blocks = 0;
    if (int rc = devb->stage1(P, h->stream, prov, nprov, cnt, dev_gbits, dev_words, &max_blocks)) return rc;
    dev = max_blocks <= kObsMax;  // longer block lists may be long tracks: host build
    lap("device stage 1 (upload, group, graph)");
  } else {
    for (int64_t i = 0; i < P->n_obs; ++i) cnt[P->obs_cam[i]] += 1.0;
    for (int64_t i = 0; i < P->n_dobs; ++i) cnt[P->dobs_cam[i]] += 1.0;
    if (int rc = allreduce_host(h, cnt.data(), nc)) return rc;
  }
  h->cam_slot_h.assign(nc, -1);
  std::vector<double> cmask((size_t)nc * 6, 0.0);
  h->ncv = 0;
  for (int i = 0; i < nc; ++i) {
    if (P->pose_const[i] || cnt[i] == 0.0) continue;
    h->cam_slot_h[i] = h->ncv++;
    for (int k = 0; k < 6; ++k) cmask[(size_t)i * 6 + k] = 1.0;
    if (i == P->gauge_axis_cam) cmask[(size_t)i * 6 + 3] = 0.0;
  }
  const int ncv_real = h->ncv;
  h->n_user = 6 * ncv_real;
  h->n = 6 * h->ncv;
  h->nt = (h->n + 31) / 32;
  std::vector<int32_t>& slot = h->cam_slot_h;  // the caller's order for now; re-assigned below from the camera graph
  // Records a DENSE chunk may hold.  A very small problem (a local bundle adjustment of a few cameras: a handful of full chunks) is
  // cut into ~40 smaller ones: every sweep costs ONE workgroup's latency, which grows with the chunk's landmarks — measured in the
  // single launch of local_lm.hip: 3 cameras / 300 landmarks 39 -> 34 us per iteration with 24 chunks instead of 6; beyond ~40
  // chunks its grid barriers (~35 ns per workgroup each) take back what the sweep gains.  MPSFM_CHUNK_RECORDS overrides.
  // landmarks of a dense chunk by the size of its camera set (dense_pts_cap): small problems, MPSFM_CHUNK_PTS_BY_CAMS overrides
  const int pts_by_cams = [&] { const char* e = std::getenv("MPSFM_CHUNK_PTS_BY_CAMS"); return e ? (std::atoi(e) != 0 ? 1 : 0) : ((!sharded(h) && h->ncv >= 1 && h->ncv <= kLocalCams) ? 1 : 0); }();
  int rec_cap = kObsMax;
  constexpr int64_t kSmallChunks = 40;
  if (const char* e = std::getenv("MPSFM_CHUNK_RECORDS")) rec_cap = std::min(std::max(std::atoi(e), 16), (int)kObsMax);
  else if (!sharded(h) && h->ncv >= 1 && h->ncv <= kLocalCams && (int64_t)P->n_obs < kSmallChunks * kObsMax)
    rec_cap = (int)std::min<int64_t>(kObsMax, std::max<int64_t>(64, ((int64_t)P->n_obs + kSmallChunks - 1) / kSmallChunks));

  // -- residual blocks grouped by landmark and merged into records.  Every host thread owns a contiguous
  //    landmark range: it scans the block lists for its landmarks (counting sort), orders each landmark's
  //    blocks by camera slot and merges a reprojection and a depth block of one (camera, landmark) pair
  //    into one record; fixed blocks (constant camera and constant landmark) are kept aside.
  struct Rec { int32_t cam; int32_t slot; uint32_t flags; double u, v, d, m, a; };
  auto deff = [&](int cam, int64_t src) {
    double b = 0.0, s = 0.0;
    if (P->shift_logscale) { b = P->shift_logscale[2 * cam]; s = P->shift_logscale[2 * cam + 1]; }
    return P->dobs_depth[src] * std::exp(s) + b;
  };
  struct MergePart {
    int p0 = 0, p1 = 0, err = 0;
    std::vector<int64_t> pstart;  // block range of every landmark of the part
    HostBuf<Blk> blks;   // recycled, uninitialised blocks (HostBlockCache): a few MB per part
    HostBuf<Rec> recs;   // capacity = the part's blocks (merging only removes); nrecs filled
    size_t nrecs = 0;
    std::vector<Rec> fixed;
    std::vector<int32_t> fixed_pt;
    std::vector<int64_t> nrec_of;  // records per landmark of the range
  };
  const int mparts = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (P->n_obs + P->n_dobs) / 32768));  // starting threads only pays above ~100 k blocks
  std::vector<MergePart> mp((size_t)mparts);
  // Phase A (host threads over landmark ranges): the blocks of every landmark side by side (counting sort).
  if (!dev) run_parts(mparts, [&](int t, int nparts) {
    MergePart& M = mp[(size_t)t];
    M.p0 = (int)((int64_t)npu * t / nparts); M.p1 = (int)((int64_t)npu * (t + 1) / nparts);
    const int p0 = M.p0, np_loc = M.p1 - M.p0;
    std::vector<int64_t>& pstart = M.pstart;
    pstart.assign((size_t)np_loc + 1, 0);
    for (int64_t i = 0; i < P->n_obs; ++i) { const unsigned q = (unsigned)(P->obs_pt[i] - p0); if (q < (unsigned)np_loc) pstart[q + 1]++; }
    for (int64_t i = 0; i < P->n_dobs; ++i) { const unsigned q = (unsigned)(P->dobs_pt[i] - p0); if (q < (unsigned)np_loc) pstart[q + 1]++; }
    for (int q = 0; q < np_loc; ++q) pstart[q + 1] += pstart[q];
    M.blks.alloc((size_t)pstart[np_loc]);
    std::vector<int64_t> fill(pstart.begin(), pstart.end() - 1);
    for (int64_t i = 0; i < P->n_obs; ++i) {
      const unsigned q = (unsigned)(P->obs_pt[i] - p0);
      if (q >= (unsigned)np_loc) continue;
      M.blks[(size_t)fill[q]++] = Blk{P->obs_cam[i], 0, 0, i};
    }
    for (int64_t i = 0; i < P->n_dobs; ++i) {
      const unsigned q = (unsigned)(P->dobs_pt[i] - p0);
      if (q >= (unsigned)np_loc) continue;
      if (!(P->dobs_depth[i] > 0.0)) { M.err = 1; return; }
      M.blks[(size_t)fill[q]++] = Blk{P->dobs_cam[i], 0, 1, i};
    }
  });
  for (const MergePart& M : mp)
    if (M.err == 1) return fail(MPSFM_EINVAL, "depth prior must be positive");
  lap("group blocks by landmark (threads)");

  // -- camera order.  Up to kIndexMaxSlots variable cameras: the camera graph (who shares a variable landmark with whom,
  //    summed over the ranks) decides the slot order — nested dissection when it shortens the dependent chain of the tile
  //    factorisation (chol_plan.h) — and which 6x6 blocks of S exist.  Beyond: the caller's order and a block skyline.
  const bool use_graph = ncv_real > 0 && ncv_real <= kIndexMaxSlots && !(std::getenv("MPSFM_CHOL_GRAPH") && std::atoi(std::getenv("MPSFM_CHOL_GRAPH")) == 0);
  CamGraph graph;
  if (use_graph) {
    graph.init(ncv_real);
    std::vector<std::vector<uint64_t>> gb((size_t)mparts);
    if (dev) {
      // the device's graph speaks provisional slots (all non-constant cameras); cameras without blocks have no slot and no edges
      std::vector<int32_t> nat_of_prov((size_t)std::max(nprov, 1), -1);
      for (int i = 0; i < nc; ++i) if (prov[(size_t)i] >= 0) nat_of_prov[(size_t)prov[(size_t)i]] = slot[(size_t)i];
      for (int a = 0; a < nprov; ++a) {
        const int na = nat_of_prov[(size_t)a];
        for (int w = 0; w < dev_words; ++w) {
          uint64_t m = dev_gbits[(size_t)a * dev_words + w];
          while (m) {
            const int b = w * 64 + __builtin_ctzll(m);
            m &= m - 1;
            const int nb = nat_of_prov[(size_t)b];
            if (na >= 0 && nb >= 0) graph.set(na, nb);
          }
        }
      }
    } else run_parts(mparts, [&](int t, int) {
      const MergePart& M = mp[(size_t)t];
      std::vector<uint64_t>& B = gb[(size_t)t];
      B.assign(graph.bits.size(), 0);
      std::vector<int32_t> sl;
      for (int q = 0; q < M.p1 - M.p0; ++q) {
        if (P->pt_const[M.p0 + q]) continue;
        sl.clear();
        for (int64_t r = M.pstart[(size_t)q]; r < M.pstart[(size_t)q + 1]; ++r) {
          const int sc = slot[(size_t)M.blks[(size_t)r].cam];
          if (sc >= 0 && std::find(sl.begin(), sl.end(), sc) == sl.end()) sl.push_back(sc);  // a handful of cameras per landmark
        }
        for (size_t a = 0; a < sl.size(); ++a)
          for (size_t b = a + 1; b < sl.size(); ++b) {
            B[(size_t)sl[a] * graph.words + (sl[b] >> 6)] |= 1ull << (sl[b] & 63);
            B[(size_t)sl[b] * graph.words + (sl[a] >> 6)] |= 1ull << (sl[a] & 63);
          }
      }
    });
    if (!dev) for (const auto& B : gb) for (size_t w = 0; w < B.size(); ++w) graph.bits[w] |= B[w];
    if (sharded(h)) {
      // union over the ranks through the sum exchange (pack_graph / unpack_graph); the number of ranks comes from the
      // exchange itself (a hook may come without world_size)
      double ones = 1.0;
      if (int rc = allreduce_host(h, &ones, 1)) return rc;
      const int world = std::max((int)std::llround(ones), 1);
      std::vector<double> packed;
      pack_graph(graph, world, packed);
      if (int rc = allreduce_host(h, packed.data(), (int64_t)packed.size())) return rc;
      unpack_graph(packed, world, graph);
    }
    lap("camera graph");
    int forced_depth = -2;
    if (const char* e = std::getenv("MPSFM_CHOL_ND")) forced_depth = std::atoi(e);  // -1: caller's order, >= 0: dissection depth
    int inv_rows = 2;
    if (const char* e = std::getenv("MPSFM_CHOL_INVERSE")) if (std::atoi(e) == 0) inv_rows = -1;
    plan_auto(graph, forced_depth, forced_depth >= -1, inv_rows < 0 ? 0 : dense_plain_max_tiles(), dense_inv_rows(), h->plan,
              [](int n, void (*fn)(void*, int), void* ctx) { run_parts(n, [&](int t, int) { fn(ctx, t); }); });
    lap("camera order + factorisation plan");
    h->nat_slot = h->plan.slot_of_nat;
    h->ncv = h->plan.nslots;                 // a permutation of the variable cameras
    h->n = h->plan.n;                        // columns of the reduced system incl. the alignment padding
    h->nt = (h->n + 31) / 32;
    for (int i = 0; i < nc; ++i)
      if (slot[(size_t)i] >= 0) slot[(size_t)i] = h->plan.slot_of_nat[(size_t)slot[(size_t)i]];
  } else {
    h->nat_slot.resize((size_t)ncv_real);
    for (int i = 0; i < ncv_real; ++i) h->nat_slot[(size_t)i] = i;
  }

  // tables both builds hand to the tail of this function
  std::vector<ChunkHdr> chunks;
  std::vector<int32_t> chunk_cams;
  HostBuf<int32_t> rec_cam, rec_pt;
  std::vector<int32_t> pt_rec_start;
  std::vector<uint32_t> blk_desc, ents;          // Schur pairs grouped by destination block, per chunk
  std::vector<int32_t> blk_ent_start;
  HostBuf<uint32_t> rec_meta;
  std::vector<uint16_t> pt_kv;
  HostBuf<double> rec_xy, rec_d, rec_m, rec_a;
  int64_t nrec_total = 0;
  int64_t nblk_reduced = 0;
  double nvarpts = 0;
  std::vector<LongHdr> lhdr;
  int64_t wl_rows = 0;
  std::vector<int32_t> fx_cam, fx_pt; std::vector<uint32_t> fx_meta; std::vector<double> fx_xy, fx_d, fx_m, fx_a;
  if (dev) {
    const bool dense_on = !(std::getenv("MPSFM_SWEEP_DENSE") && std::atoi(std::getenv("MPSFM_SWEEP_DENSE")) == 0);
    const int rc2 = devb->stage2(slot, dense_on, rec_cap, pts_by_cams, DB);
    if (rc2 < 0) return rc2;
    if (rc2 == MPSFM_DEVBUILD_FALLBACK) {
      // long tracks: the host phases run after all — Phase A first, which was skipped
      DB.release();
      dev = false;
      run_parts(mparts, [&](int t, int nparts) {
        MergePart& M = mp[(size_t)t];
        M.p0 = (int)((int64_t)npu * t / nparts); M.p1 = (int)((int64_t)npu * (t + 1) / nparts);
        const int p0 = M.p0, np_loc = M.p1 - M.p0;
        std::vector<int64_t>& pstart = M.pstart;
        pstart.assign((size_t)np_loc + 1, 0);
        for (int64_t i = 0; i < P->n_obs; ++i) { const unsigned q = (unsigned)(P->obs_pt[i] - p0); if (q < (unsigned)np_loc) pstart[q + 1]++; }
        for (int64_t i = 0; i < P->n_dobs; ++i) { const unsigned q = (unsigned)(P->dobs_pt[i] - p0); if (q < (unsigned)np_loc) pstart[q + 1]++; }
        for (int q = 0; q < np_loc; ++q) pstart[q + 1] += pstart[q];
        M.blks.alloc((size_t)pstart[np_loc]);
        std::vector<int64_t> fill(pstart.begin(), pstart.end() - 1);
        for (int64_t i = 0; i < P->n_obs; ++i) {
          const unsigned q = (unsigned)(P->obs_pt[i] - p0);
          if (q >= (unsigned)np_loc) continue;
          M.blks[(size_t)fill[q]++] = Blk{P->obs_cam[i], 0, 0, i};
        }
        for (int64_t i = 0; i < P->n_dobs; ++i) {
          const unsigned q = (unsigned)(P->dobs_pt[i] - p0);
          if (q >= (unsigned)np_loc) continue;
          M.blks[(size_t)fill[q]++] = Blk{P->dobs_cam[i], 0, 1, i};  // depths were validated by stage 1
        }
      });
      lap("device build not applicable: host phases");
    } else {
      chunks.swap(DB.chunks); chunk_cams.swap(DB.chunk_cams);
      h->perm.swap(DB.order);
      h->np = DB.np; h->np_chunked = DB.np_chunked; h->nfixed = DB.nfixed;
      h->nblocks_total = P->n_obs + P->n_dobs;
      nrec_total = DB.nrec; nblk_reduced = DB.nblk_reduced; nvarpts = DB.nvarpts;
      if (nrec_total > (int64_t)INT32_MAX) return fail(MPSFM_EUNSUPPORTED, "more than 2^31 records on one device");
      h->nchunks = (int)chunks.size();
      h->nlong = 0; h->nrec = nrec_total; h->nblocks_reduced = nblk_reduced;
      h->nblocks_global = (double)h->nblocks_total; h->nblocks_reduced_global = (double)nblk_reduced; h->nvarpts_global = nvarpts;
      lap("device stage 2 (order, chunks, records)");
      // pair tables: a sentinel per dense chunk; the general chunks (landmarks with more than kDenseCams cameras or two records of one
      // camera; they come last) get theirs from the host, which needs their record words and landmark tables back
      size_t g0 = 0;
      while (g0 < chunks.size() && chunks[g0].dense) ++g0;
      blk_ent_start.assign(g0, 0);
      if (g0 < chunks.size()) {
        const int64_t r0 = chunks[g0].rec0, k0 = chunks[g0].pt0, nrg = nrec_total - r0, nkg = h->np_chunked - k0;
        std::vector<uint32_t> rm((size_t)std::max<int64_t>(nrg, 1));
        std::vector<uint16_t> kvs((size_t)std::max<int64_t>(nkg, 1));
        std::vector<int32_t> prs((size_t)std::max<int64_t>(nkg, 1));
        HIP_TRY(hipMemcpyAsync(rm.data(), DB.d_rec_meta + r0, 4 * (size_t)nrg, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(kvs.data(), DB.d_pt_kv + k0, 2 * (size_t)nkg, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(prs.data(), DB.d_pt_rec_start + k0, 4 * (size_t)nkg, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        PairScratch ps;
        for (size_t c = g0; c < chunks.size(); ++c) {
          if (chunks[c].dense) return fail(MPSFM_EUNSUPPORTED, "internal: dense chunks must precede the general ones");
          append_pair_tables(chunks[c], rm.data() - r0, kvs.data() - k0, prs.data() - k0, h->perm.data(), P->pt_const, blk_desc, ents, blk_ent_start, ps);
        }
        if (ents.size() > (size_t)INT32_MAX) return fail(MPSFM_EUNSUPPORTED, "too many Schur pairs for 32-bit entry offsets");
        lap("pair tables of the general chunks (host)");
      }
    }
  }
  if (!dev) {
  // Phase B (the same host threads): every landmark's blocks ordered by (final) camera slot, a reprojection and a depth
  // block of one (camera, landmark) pair merged into one record.
  run_parts(mparts, [&](int t, int) {
    MergePart& M = mp[(size_t)t];
    const int p0 = M.p0, np_loc = M.p1 - M.p0;
    Blk* const blks = M.blks.data();
    for (size_t q = 0; q < M.blks.size(); ++q) blks[q].key = slot[(size_t)blks[q].cam] < 0 ? INT32_MAX : slot[(size_t)blks[q].cam];
    M.recs.alloc(M.blks.size());
    M.nrecs = 0;
    M.nrec_of.assign((size_t)np_loc, 0);
    for (int q = 0; q < np_loc; ++q) {
      const int p = p0 + q;
      Blk* const b0 = blks + M.pstart[(size_t)q]; Blk* const b1 = blks + M.pstart[(size_t)q + 1];
      std::sort(b0, b1, [](const Blk& x, const Blk& y) {
        if (x.key != y.key) return x.key < y.key;
        if (x.cam != y.cam) return x.cam < y.cam;
        if (x.kind != y.kind) return x.kind < y.kind;
        return x.src < y.src;
      });
      const size_t before = M.nrecs;
      for (auto it = b0; it != b1;) {
        auto je = it;
        while (je != b1 && je->cam == it->cam) ++je;
        auto mid = it;
        while (mid != je && mid->kind == 0) ++mid;
        const int64_t nr = mid - it, nd = je - mid;
        const bool is_fixed = (slot[it->cam] < 0) && P->pt_const[p];
        for (int64_t k = 0; k < std::max(nr, nd); ++k) {
          Rec r{it->cam, slot[it->cam], 0, 0, 0, 1.0, 0.0, 1.0};
          if (k < nr) { const int64_t s = (it + k)->src; r.flags |= kRecHasReproj; r.u = P->obs_xy[2 * s]; r.v = P->obs_xy[2 * s + 1]; }
          if (k < nd) {
            const int64_t s = (mid + k)->src;
            r.flags |= kRecHasDepth; r.d = deff(it->cam, s); r.m = P->dobs_magnitude[s]; r.a = P->dobs_param[s];
            if (!(r.d > 0.0)) { M.err = 2; return; }
            r.d = std::log(r.d);  // the residual is log Z - log d: the records carry log d (one logarithm less per evaluation)
          }
          if (is_fixed) { M.fixed.push_back(r); M.fixed_pt.push_back(p); }
          else M.recs[M.nrecs++] = r;
        }
        it = je;
      }
      M.nrec_of[(size_t)q] = (int64_t)(M.nrecs - before);
    }
    M.blks.release();
  });
  for (const MergePart& M : mp)
    if (M.err == 2) return fail(MPSFM_EINVAL, "shifted/scaled depth prior must be positive");
  lap("sort + merge into records (threads)");
  std::vector<int64_t> prec((size_t)npu + 1, 0);  // record range per caller landmark
  HostBuf<Rec> recs;  // uninitialised: value-initialising tens of MB on one thread costs milliseconds
  std::vector<Rec> fixed;
  std::vector<int32_t> fixed_pt;
  {
    std::vector<int64_t> base((size_t)mparts + 1, 0);
    for (int t = 0; t < mparts; ++t) base[(size_t)t + 1] = base[(size_t)t] + (int64_t)mp[(size_t)t].nrecs;
    recs.alloc((size_t)base[(size_t)mparts]);
    run_parts(mparts, [&](int t, int) {
      MergePart& M = mp[(size_t)t];
      std::copy(M.recs.data(), M.recs.data() + M.nrecs, recs.data() + base[(size_t)t]);
      int64_t o = base[(size_t)t];
      for (int q = 0; q < M.p1 - M.p0; ++q) { prec[(size_t)(M.p0 + q)] = o; o += M.nrec_of[(size_t)q]; }
      M.recs.release();
    });
    prec[(size_t)npu] = base[(size_t)mparts];
    for (MergePart& M : mp) {
      fixed.insert(fixed.end(), M.fixed.begin(), M.fixed.end());
      fixed_pt.insert(fixed_pt.end(), M.fixed_pt.begin(), M.fixed_pt.end());
    }
  }
  h->nfixed = (int64_t)fixed.size();
  h->nblocks_total = P->n_obs + P->n_dobs;
  lap("merge records");
  // -- landmark order: those with records sorted by their camera-slot list, then the rest that
  //    are referenced by fixed blocks only
  std::vector<int32_t> order; order.reserve(npu);
  for (int p = 0; p < npu; ++p) if (prec[p + 1] > prec[p]) order.push_back(p);
  {
    // sort key: the first six camera slots of the track (16 bits each; slots beyond 65534 and constant
    // cameras saturate), then the track length, then the landmark index — neighbours in this order share
    // cameras, which keeps the set of S blocks a chunk touches small
    struct Key { uint64_t k1, k2; int32_t p; };
    std::vector<Key> keyed(order.size());
    auto key_less = [](const Key& a, const Key& b) {
      if (a.k1 != b.k1) return a.k1 < b.k1;
      if (a.k2 != b.k2) return a.k2 < b.k2;
      return a.p < b.p;
    };
    // sorted runs per thread, then pairwise merges (the order is total, so the result does not depend on the split)
    const int sparts = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), order.size() / 8192));
    std::vector<size_t> cut((size_t)sparts + 1);
    for (int t = 0; t <= sparts; ++t) cut[(size_t)t] = order.size() * (size_t)t / (size_t)sparts;
    run_parts(sparts, [&](int t, int) {
      for (size_t q = cut[(size_t)t]; q < cut[(size_t)t + 1]; ++q) {
        const int pnt = order[q];
        const int64_t n_p = prec[pnt + 1] - prec[pnt];
        uint64_t key[2] = {0, 0};
        for (int64_t k = 0; k < 6; ++k) {
          uint64_t sk = 0xffff;
          if (k < n_p && recs[prec[pnt] + k].slot >= 0) sk = (uint64_t)std::min(recs[prec[pnt] + k].slot, 0xfffe);
          key[k / 4] = (key[k / 4] << 16) | sk;
        }
        key[1] = (key[1] << 32) | (uint64_t)std::min<int64_t>(n_p, 0xffffffff);
        keyed[q] = Key{key[0], key[1], pnt};
      }
      std::sort(keyed.begin() + (std::ptrdiff_t)cut[(size_t)t], keyed.begin() + (std::ptrdiff_t)cut[(size_t)t + 1], key_less);
    });
    for (int width = 1; width < sparts; width *= 2) {
      std::vector<int> lefts;
      for (int t = 0; t + width < sparts; t += 2 * width) lefts.push_back(t);
      run_parts((int)lefts.size(), [&](int j, int) {
        const int t = lefts[(size_t)j];
        std::inplace_merge(keyed.begin() + (std::ptrdiff_t)cut[(size_t)t], keyed.begin() + (std::ptrdiff_t)cut[(size_t)(t + width)],
                           keyed.begin() + (std::ptrdiff_t)cut[(size_t)std::min(t + 2 * width, sparts)], key_less);
      });
    }
    for (size_t q = 0; q < order.size(); ++q) order[q] = keyed[q].p;
  }
  lap("sort landmarks by key");
  // landmarks whose track does not fit one chunk are swept by a workgroup of their own
  auto is_long = [&](int p) {
    const int64_t r_p = prec[p + 1] - prec[p];
    if (r_p > kObsMax) return true;
    int distinct = 0, last = -2;
    for (int64_t r = prec[p]; r < prec[p + 1]; ++r)
      if (recs[r].slot >= 0 && recs[r].slot != last) { ++distinct; last = recs[r].slot; }
    return distinct > kLocalCamsMax;
  };
  std::vector<uint8_t> long_flag((size_t)npu + 1, 0);
  parallel_ranges((int64_t)order.size(), 8192, [&](int64_t q0, int64_t q1) {
    for (int64_t q = q0; q < q1; ++q) long_flag[(size_t)order[(size_t)q]] = is_long(order[(size_t)q]) ? 1 : 0;
  });
  auto first_long = std::stable_partition(order.begin(), order.end(), [&](int p) { return !long_flag[(size_t)p]; });
  h->np_chunked = (int64_t)(first_long - order.begin());
  // Landmarks the dense sweep cannot take — more than kDenseCams variable cameras, or two records of one camera for a variable
  // landmark — go behind the others (same relative order), so that they form chunks of their own for the general kernel and
  // every other chunk is dense by construction (the cut below keeps those within kDenseCams cameras / kDensePts landmarks).
  const bool dense_on = !(std::getenv("MPSFM_SWEEP_DENSE") && std::atoi(std::getenv("MPSFM_SWEEP_DENSE")) == 0);
  std::vector<uint8_t> heavy_flag((size_t)npu + 1, dense_on ? 0 : 1);
  if (dense_on) {
    parallel_ranges(h->np_chunked, 8192, [&](int64_t q0, int64_t q1) {
      for (int64_t q = q0; q < q1; ++q) {
        const int p = order[(size_t)q];
        int distinct = 0, last = -2;
        bool dup = false;
        for (int64_t r = prec[p]; r < prec[p + 1]; ++r) {
          if (recs[r].slot < 0) continue;
          if (recs[r].slot != last) { ++distinct; last = recs[r].slot; }
          else dup = true;  // records are slot-sorted: two of one camera are neighbours
        }
        heavy_flag[(size_t)p] = (distinct > kDenseCams || (dup && !P->pt_const[p])) ? 1 : 0;
      }
    });
    std::stable_partition(order.begin(), first_long, [&](int p) { return !heavy_flag[(size_t)p]; });
  }
  const int64_t n_long = (int64_t)(order.end() - first_long);
  {
    std::vector<uint8_t> seen((size_t)npu + 1, 0);
    for (int p : order) seen[p] = 1;
    for (int32_t p : fixed_pt) if (!seen[p]) { seen[p] = 1; order.push_back(p); }
  }
  h->np = (int64_t)order.size();
  h->perm = order;
  std::vector<int32_t> inv((size_t)npu + 1, -1);
  for (int64_t k = 0; k < h->np; ++k) inv[order[k]] = (int32_t)k;

  lap("order landmarks");
  // -- chunking.  Pass 1 (sequential, greedy): cut the ordered landmarks into chunks and collect each chunk's
  //    sorted camera slots.  Pass 2 (host threads over contiguous chunk ranges): records, local camera
  //    indices and the block-major Schur pair tables of every chunk.
  pt_rec_start.assign((size_t)h->np + 1, 0);
  pt_kv.assign((size_t)h->np + 1, 0xffff);
  {
    std::vector<int64_t> rec_off((size_t)h->np_chunked + 1, 0);  // first record of every chunked landmark
    for (int64_t k = 0; k < h->np_chunked; ++k) rec_off[(size_t)k + 1] = rec_off[(size_t)k] + (prec[order[k] + 1] - prec[order[k]]);
    {
      // The greedy cut is sequential by nature; the ordered landmarks are therefore split into a FIXED number
      // of segments (independent of the thread count, so the tables are the same on every machine), each cut
      // greedily on its own with a forced chunk boundary at the segment ends.
      struct Seg { std::vector<ChunkHdr> chunks; std::vector<int32_t> cams; };
      const int nseg = (int)std::max<int64_t>(1, std::min<int64_t>(64, h->np_chunked / 4096));
      std::vector<Seg> segs((size_t)nseg);
      auto cut_segment = [&](int sidx) {
        Seg& G = segs[(size_t)sidx];
        const int64_t k0 = h->np_chunked * sidx / nseg, k1 = h->np_chunked * (sidx + 1) / nseg;
        std::vector<int32_t> cur_cams, pc, uni;  // sorted slots of the open chunk
        int64_t c_first = k0, c_nrec = 0;
        auto close_chunk = [&](int64_t end_pt) {
          if (end_pt == c_first) return;
          ChunkHdr H{};
          H.rec0 = (int32_t)rec_off[(size_t)c_first]; H.nrec = (int32_t)(rec_off[(size_t)end_pt] - rec_off[(size_t)c_first]);
          H.pt0 = (int32_t)c_first; H.npt = (int32_t)(end_pt - c_first);
          H.cam0 = (int32_t)G.cams.size(); H.ncam = (int32_t)cur_cams.size();  // segment-local for now
          G.cams.insert(G.cams.end(), cur_cams.begin(), cur_cams.end());
          G.chunks.push_back(H);
          cur_cams.clear(); c_first = end_pt; c_nrec = 0;
        };
        for (int64_t k = k0; k < k1; ++k) {
          const int p = order[k];
          const int64_t r_p = prec[p + 1] - prec[p];
          pc.clear();
          for (int64_t r = prec[p]; r < prec[p + 1]; ++r) if (recs[r].slot >= 0) pc.push_back(recs[r].slot);
          pc.erase(std::unique(pc.begin(), pc.end()), pc.end());  // records are slot-sorted
          const bool subset = std::includes(cur_cams.begin(), cur_cams.end(), pc.begin(), pc.end());
          size_t nuni = cur_cams.size();
          if (!subset) {
            uni.clear();
            std::set_union(cur_cams.begin(), cur_cams.end(), pc.begin(), pc.end(), std::back_inserter(uni));
            nuni = uni.size();
          }
          const bool hv = heavy_flag[(size_t)p] != 0;
          const bool too_big = (c_nrec + r_p > (hv ? kObsMax : rec_cap)) || (k - c_first + 1 > (hv ? kPtsMax : dense_pts_cap((int)nuni, pts_by_cams))) || ((int)nuni > (hv ? kLocalCamsMax : kDenseCams)) ||
                               (hv != (heavy_flag[(size_t)order[(size_t)c_first]] != 0));  // dense and general landmarks never share a chunk
          if (k > c_first && too_big) {
            close_chunk(k);
            cur_cams = pc;
          } else if (!subset) {
            cur_cams.swap(uni);
          }
          c_nrec += r_p;
        }
        close_chunk(k1);
      };
      run_parts(std::min(host_threads(), nseg), [&](int t, int nparts) { for (int sidx = t; sidx < nseg; sidx += nparts) cut_segment(sidx); });
      for (Seg& G : segs) {
        const int32_t cbase = (int32_t)chunk_cams.size();
        for (ChunkHdr& H : G.chunks) { H.cam0 += cbase; chunks.push_back(H); }
        chunk_cams.insert(chunk_cams.end(), G.cams.begin(), G.cams.end());
      }
    }
    lap("chunk boundaries");
    const size_t nrec_chunked = (size_t)rec_off[(size_t)h->np_chunked];
    nrec_total = (int64_t)nrec_chunked;
    for (int64_t k = h->np_chunked; k < h->np_chunked + n_long; ++k) nrec_total += prec[order[k] + 1] - prec[order[k]];
    if (nrec_total > (int64_t)INT32_MAX) return fail(MPSFM_EUNSUPPORTED, "more than 2^31 records on one device");
    const size_t nr = (size_t)nrec_total;
    rec_cam.alloc(nr); rec_pt.alloc(nr); rec_meta.alloc(nr); rec_xy.alloc(2 * nr); rec_d.alloc(nr); rec_m.alloc(nr); rec_a.alloc(nr);
    lap("size record arrays");
    struct ChunkPart {
      std::vector<uint32_t> blk_desc, ents;
      std::vector<int32_t> blk_ent_start;
      int64_t nblk_reduced = 0;
      double nvarpts = 0;
    };
    const int nch = (int)chunks.size();
    const int cparts = std::max(1, std::min(host_threads(), nch / 48));
    std::vector<ChunkPart> cp((size_t)cparts);
    run_parts(cparts, [&](int t, int nparts) {
      ChunkPart& C = cp[(size_t)t];
      PairScratch ps;
      for (int c = (int)((int64_t)nch * t / nparts); c < (int)((int64_t)nch * (t + 1) / nparts); ++c) {
        ChunkHdr& H = chunks[(size_t)c];
        const int32_t* cams = chunk_cams.data() + H.cam0;
        const int64_t c_first = H.pt0, end_pt = (int64_t)H.pt0 + H.npt;
        bool dup = false;    // two records of one camera for one variable landmark
        int64_t w = H.rec0;  // next record
        for (int64_t k = c_first; k < end_pt; ++k) {
          const int p = order[k];
          pt_rec_start[(size_t)k] = (int32_t)w;
          const int rbase = (int)(w - H.rec0);  // chunk-relative index of this landmark's first record
          int kv = 0;
          for (int64_t r = prec[p]; r < prec[p + 1]; ++r, ++w) {
            const Rec& R = recs[r];
            uint32_t lcam = kLcamConst;
            if (R.slot >= 0) {
              lcam = (uint32_t)(std::lower_bound(cams, cams + H.ncam, R.slot) - cams);
              ++kv;
            }
            rec_cam[(size_t)w] = R.cam; rec_pt[(size_t)w] = (int32_t)k;
            rec_meta[(size_t)w] = lcam | ((uint32_t)(k - c_first) << 8) | R.flags;
            rec_xy[2 * (size_t)w] = R.u; rec_xy[2 * (size_t)w + 1] = R.v; rec_d[(size_t)w] = R.d; rec_m[(size_t)w] = R.m; rec_a[(size_t)w] = R.a;
            C.nblk_reduced += ((R.flags & kRecHasReproj) ? 1 : 0) + ((R.flags & kRecHasDepth) ? 1 : 0);
          }
          if (!P->pt_const[p]) {
            pt_kv[(size_t)k] = (uint16_t)kv;
            C.nvarpts += 1;
            // two records of one camera for this landmark (records are slot-sorted: they are neighbours)
            for (int i = 1; i < kv; ++i)
              if ((rec_meta[(size_t)H.rec0 + rbase + i] & 0xff) == (rec_meta[(size_t)H.rec0 + rbase + i - 1] & 0xff)) dup = true;
          }
        }
        // Chunks that form their Schur blocks as one dense product (k_track_sweep) need no pair tables at all.
        H.dense = heavy_flag[(size_t)order[(size_t)c_first]] ? 0 : 1;  // by construction: <= kDenseCams cameras, <= kDensePts landmarks, no duplicates
        H.slab0 = 0;
        (void)dup;
        append_pair_tables(H, rec_meta.data(), pt_kv.data(), pt_rec_start.data(), order.data(), P->pt_const, C.blk_desc, C.ents, C.blk_ent_start, ps);
      }
    });
    lap("chunk records + pairs (threads)");
    // concatenate the per-thread pair tables and make the chunk offsets global
    std::vector<size_t> bbase((size_t)cparts + 1, 0), ebase((size_t)cparts + 1, 0), sbase((size_t)cparts + 1, 0);
    for (int t = 0; t < cparts; ++t) {
      bbase[(size_t)t + 1] = bbase[(size_t)t] + cp[(size_t)t].blk_desc.size();
      ebase[(size_t)t + 1] = ebase[(size_t)t] + cp[(size_t)t].ents.size();
      sbase[(size_t)t + 1] = sbase[(size_t)t] + cp[(size_t)t].blk_ent_start.size();
      nblk_reduced += cp[(size_t)t].nblk_reduced; nvarpts += cp[(size_t)t].nvarpts;
    }
    if (ebase[(size_t)cparts] > (size_t)INT32_MAX) return fail(MPSFM_EUNSUPPORTED, "too many Schur pairs for 32-bit entry offsets");
    blk_desc.resize(bbase[(size_t)cparts]); ents.resize(ebase[(size_t)cparts]); blk_ent_start.resize(sbase[(size_t)cparts]);
    run_parts(cparts, [&](int t, int nparts) {
      const ChunkPart& C = cp[(size_t)t];
      std::copy(C.blk_desc.begin(), C.blk_desc.end(), blk_desc.begin() + (std::ptrdiff_t)bbase[(size_t)t]);
      std::copy(C.ents.begin(), C.ents.end(), ents.begin() + (std::ptrdiff_t)ebase[(size_t)t]);
      std::copy(C.blk_ent_start.begin(), C.blk_ent_start.end(), blk_ent_start.begin() + (std::ptrdiff_t)sbase[(size_t)t]);
      for (int c = (int)((int64_t)nch * t / nparts); c < (int)((int64_t)nch * (t + 1) / nparts); ++c) {
        chunks[(size_t)c].blk0 += (int32_t)bbase[(size_t)t];
        chunks[(size_t)c].ent0 += (int32_t)ebase[(size_t)t];
      }
    });
  }
  lap("concatenate pair tables");
  if (ents.size() > (size_t)INT32_MAX) return fail(MPSFM_EUNSUPPORTED, "too many Schur pairs for 32-bit entry offsets");
  h->nchunks = (int)chunks.size();
  {
    size_t w = h->np_chunked > 0 ? (size_t)(pt_rec_start[(size_t)h->np_chunked - 1] + (prec[order[h->np_chunked - 1] + 1] - prec[order[h->np_chunked - 1]])) : 0;
    for (int64_t k = h->np_chunked; k < h->np_chunked + n_long; ++k) {
      const int p = order[k];
      LongHdr L{};
      L.rec0 = (int32_t)w; L.pt = (int32_t)k; L.w0 = wl_rows;
      pt_rec_start[k] = L.rec0;
      int kv = 0;
      for (int64_t r = prec[p]; r < prec[p + 1]; ++r, ++w) {
        const Rec& R = recs[r];
        if (R.slot >= 0) ++kv;
        rec_cam[w] = R.cam; rec_pt[w] = (int32_t)k;
        rec_meta[w] = (R.slot >= 0 ? 0u : kLcamConst) | R.flags;
        rec_xy[2 * w] = R.u; rec_xy[2 * w + 1] = R.v; rec_d[w] = R.d; rec_m[w] = R.m; rec_a[w] = R.a;
        nblk_reduced += ((R.flags & kRecHasReproj) ? 1 : 0) + ((R.flags & kRecHasDepth) ? 1 : 0);
      }
      L.nrec = (int32_t)w - L.rec0;
      L.kv = P->pt_const[p] ? 0 : kv;
      if (!P->pt_const[p]) { pt_kv[k] = (uint16_t)std::min(kv, 0xfffe); nvarpts += 1; }
      wl_rows += kv;
      lhdr.push_back(L);
    }
  }
  h->nlong = (int)lhdr.size();
  h->nrec = nrec_total;
  h->nblocks_reduced = nblk_reduced;
  for (int64_t k = h->np_chunked + n_long; k <= h->np; ++k) pt_rec_start[k] = (int32_t)h->nrec;
  {
    double tot[3] = {(double)h->nblocks_total, (double)nblk_reduced, nvarpts};
    if (int rc = allreduce_host(h, tot, 3)) return rc;
    h->nblocks_global = tot[0]; h->nblocks_reduced_global = tot[1]; h->nvarpts_global = tot[2];
  }
  // -- fixed records (landmark index re-ordered)
  for (size_t i = 0; i < fixed.size(); ++i) {
    fx_cam.push_back(fixed[i].cam); fx_pt.push_back(inv[fixed_pt[i]]); fx_meta.push_back(fixed[i].flags);
    fx_xy.push_back(fixed[i].u); fx_xy.push_back(fixed[i].v); fx_d.push_back(fixed[i].d); fx_m.push_back(fixed[i].m); fx_a.push_back(fixed[i].a);
  }
  }  // host phases

  lap("chunks + pair tables");
  if (h->opt.verbose >= 2 && !chunks.empty()) {
    double sr = 0, sp = 0, sc = 0, sb = 0, se = 0, sd = 0; int mb = 0, mc = 0;
    for (const ChunkHdr& H : chunks) { sr += H.nrec; sp += H.npt; sc += H.ncam; sb += H.nblk; se += H.nent; sd += H.dense; mb = std::max(mb, H.nblk); mc = std::max(mc, H.ncam); }
    const double n = (double)chunks.size();
    std::fprintf(stderr, "[mpsfm_ba] build: %zu chunks; per chunk: %.1f records, %.1f landmarks, %.1f cameras (max %d), %.1f work items (max %d), %.1f pairs; %.0f %% of the chunks take the dense product\n",
                 chunks.size(), sr / n, sp / n, sc / n, mc, sb / n, mb, se / n, 100.0 * sd / n);
    int hist[kDenseCams + 2] = {0};
    for (const ChunkHdr& H : chunks) ++hist[std::min<int>(H.ncam, kDenseCams + 1)];
    std::fprintf(stderr, "[mpsfm_ba] build: chunks by number of variable cameras:");
    for (int c = 0; c <= kDenseCams + 1; ++c) std::fprintf(stderr, " %s%d: %d", c > kDenseCams ? ">" : "", c > kDenseCams ? kDenseCams : c, hist[c]);
    std::fprintf(stderr, "\n");
  }
  // -- which 6x6 blocks of S exist, and the tables of the dense factorisation
  if (use_graph) {
    const int ns = h->ncv;
    h->sky_index.assign((size_t)ns * (size_t)ns, -1);
    int32_t nblk = 0;
    for (int sj = 0; sj < ns; ++sj) {
      const int j = h->plan.nat_of_slot[(size_t)sj];
      if (j < 0) continue;
      for (int si = 0; si <= sj; ++si) {
        const int i = h->plan.nat_of_slot[(size_t)si];
        if (i < 0) continue;
        if (i == j || graph.get(i, j)) h->sky_index[(size_t)sj * ns + si] = nblk++;
      }
    }
    h->sblk_blocks = nblk;
    if (int rc2 = dev_upload(&h->d_sky_index, h->sky_index)) return rc2;
    if (int rc2 = upload_plan(h, nblk)) return rc2;
  } else {
    // block skyline (DenseEnvelope): which 6x6 blocks of S can be nonzero follows from the static Schur pair tables;
    // first_blk[c] = lowest camera slot that shares a landmark with slot c
    const int ncv = h->ncv, nt = h->nt, n = h->n;
    std::vector<int32_t> first_blk((size_t)std::max(ncv, 1));
    for (int c = 0; c < ncv; ++c) first_blk[(size_t)c] = c;
    for (const ChunkHdr& H : chunks) {
      const int32_t* cams = chunk_cams.data() + H.cam0;
      if (H.dense)  // no pair table: every pair of the chunk's (sorted) cameras may be coupled
        for (int q = 1; q < H.ncam; ++q) first_blk[(size_t)cams[q]] = std::min(first_blk[(size_t)cams[q]], cams[0]);
      for (int b = 0; b < H.nblk; ++b) {
        const uint32_t d = blk_desc[(size_t)H.blk0 + (size_t)b];
        const int si = cams[d & 0xff], sj = cams[(d >> 8) & 0xff];
        const int lo = std::min(si, sj), hi = std::max(si, sj);
        first_blk[(size_t)hi] = std::min(first_blk[(size_t)hi], lo);
      }
    }
    for (const LongHdr& L : lhdr) {  // a long track couples all its variable cameras
      int lo = INT32_MAX;
      for (int r = 0; r < L.kv; ++r) lo = std::min(lo, slot[rec_cam[(size_t)L.rec0 + (size_t)r]]);
      for (int r = 0; r < L.kv; ++r) { int32_t& f = first_blk[(size_t)slot[rec_cam[(size_t)L.rec0 + (size_t)r]]]; f = std::min(f, lo); }
    }
    const char* envs = std::getenv("MPSFM_CHOL_ENVELOPE");
    if (envs && std::atoi(envs) == 0) std::fill(first_blk.begin(), first_blk.end(), 0);  // A/B: treat S as dense
    if (sharded(h) && ncv > 0) {
      // landmark shards see different camera pairs: every rank needs the UNION.  The hook only sums, so the minimum over
      // ranks is found by bisection on indicator sums (the same number of rounds on every rank).
      std::vector<double> lo((size_t)ncv, 0.0), hi((size_t)ncv), ind((size_t)ncv);
      for (int c = 0; c < ncv; ++c) hi[(size_t)c] = (double)c;
      int rounds = 1;
      while ((1 << rounds) < ncv + 1) ++rounds;
      for (int it = 0; it < rounds; ++it) {
        for (int c = 0; c < ncv; ++c) ind[(size_t)c] = first_blk[(size_t)c] <= (int)std::floor(0.5 * (lo[(size_t)c] + hi[(size_t)c])) ? 1.0 : 0.0;
        if (int rc2 = allreduce_host(h, ind.data(), ncv)) return rc2;
        for (int c = 0; c < ncv; ++c) {
          const double mid = std::floor(0.5 * (lo[(size_t)c] + hi[(size_t)c]));
          if (ind[(size_t)c] > 0.0) hi[(size_t)c] = mid; else lo[(size_t)c] = std::min(mid + 1.0, hi[(size_t)c]);
        }
      }
      for (int c = 0; c < ncv; ++c) first_blk[(size_t)c] = (int32_t)hi[(size_t)c];
    }
    h->sky_first = first_blk;
    h->sky_start.assign((size_t)ncv + 1, 0);
    for (int c = 0; c < ncv; ++c) h->sky_start[(size_t)c + 1] = h->sky_start[(size_t)c] + (c - first_blk[(size_t)c] + 1);
    h->sblk_blocks = h->sky_start[(size_t)ncv];
    if (int rc2 = dev_upload(&h->d_sky_first, h->sky_first)) return rc2;
    if (int rc2 = dev_upload(&h->d_sky_start, h->sky_start)) return rc2;
    std::vector<int32_t> first((size_t)nt + 1, 0);
    for (int ti = 0; ti < nt; ++ti) {
      int f = ti;
      for (int r = ti * 32; r < std::min(n, ti * 32 + 32); ++r) f = std::min(f, (6 * first_blk[(size_t)(r / 6)]) / 32);
      first[(size_t)ti] = f;
    }
    first[(size_t)nt] = 0;  // the right-hand-side row
    // the factorisation plan of the skyline in the caller's order: tile (ti, tj) can be nonzero for tj >= first[ti]
    {
      std::vector<uint8_t> pat((size_t)nt * (size_t)nt, 0);
      for (int ti = 0; ti < nt; ++ti)
        for (int tj = first[(size_t)ti]; tj < ti; ++tj) pat[(size_t)ti * nt + tj] = 1;
      CholPlan& PL = h->plan;
      PL.ncv = ncv; PL.nslots = ncv; PL.n = n; PL.nd_depth = -1;
      PL.slot_of_nat = h->nat_slot; PL.nat_of_slot = h->nat_slot;  // identity, no padding: slot_of_col stays empty (NULL map)
      plan_from_pattern(pat, nt, nt <= dense_plain_max_tiles() && !(std::getenv("MPSFM_CHOL_INVERSE") && std::atoi(std::getenv("MPSFM_CHOL_INVERSE")) == 0),
                        dense_inv_rows(), PL);
      if (int rc2 = upload_plan(h, h->sblk_blocks)) return rc2;
    }
    if (h->opt.verbose >= 2) {
      int64_t inside = 0;
      for (int ti = 0; ti < nt; ++ti) inside += ti - first[(size_t)ti] + 1;
      std::fprintf(stderr, "[mpsfm_ba] build: block skyline %lld of %lld tiles\n", (long long)inside, (long long)nt * (nt + 1) / 2);
    }
  }

  // -- slabs of the dense chunks and the tables of their reduction (k_reduce_slabs): per destination — a block of S or a camera's
  //    vectors — the slab positions that contribute, in chunk order; destinations with many sources are split into parts
  std::vector<RedDest> red_dests;
  std::vector<int32_t> red_srcs, diag_block;
  bool slab_tables_on_device = false;
  int64_t slab_units = 0;
  {
    h->n_dense = 0;
    for (size_t c = 0; c < chunks.size(); ++c) {
      ChunkHdr& H = chunks[c];
      if (!H.dense) continue;
      if ((int)c != h->n_dense) return fail(MPSFM_EUNSUPPORTED, "internal: dense chunks must precede the general ones");
      H.slab0 = (int32_t)slab_units;
      slab_units += slab_doubles(H.ncam) / 18;
      if (slab_units > (int64_t)INT32_MAX) return fail(MPSFM_EUNSUPPORTED, "slabs of the dense chunks exceed 2^31 units");
      h->n_dense = (int)c + 1;
    }
    const int64_t nsb = h->sblk_blocks;
    // a device-built handle forms the tables on the device too (DevBuilder::slab_tables, behind the uploads below)
    {
      const char* e = std::getenv("MPSFM_SLAB_TABLES_HOST");  // 1: host loop, 0: device kernels (tests), unset: by size — below ~500
      const int pref = e ? std::atoi(e) : -1;                  // chunks the host loop is quicker than the launches
      slab_tables_on_device = dev && devb && h->n_dense > 0 && (pref == 0 || (pref < 0 && h->n_dense >= 512));
    }
    auto host_sky = [&](int si, int sj) -> int64_t {
      return use_graph ? (int64_t)h->sky_index[(size_t)sj * (size_t)h->ncv + (size_t)si] : h->sky_start[(size_t)sj] + (si - h->sky_first[(size_t)sj]);
    };
    const size_t ndst = (size_t)nsb + (size_t)std::max(h->ncv, 0);
    diag_block.assign((size_t)std::max(h->ncv, 1), -1);
    for (int sl = 0; sl < h->ncv; ++sl) {
      if (use_graph && h->plan.nat_of_slot[(size_t)sl] < 0) continue;
      const int64_t b = host_sky(sl, sl);
      if (b >= 0 && b < nsb) diag_block[(size_t)sl] = (int32_t)b;
    }
    std::vector<int32_t> cnt(ndst + 1, 0);
    for (int pass = 0; pass < 2 && !slab_tables_on_device; ++pass) {  // count, then place (chunk order within a destination)
      if (pass == 1) {
        for (size_t d = 1; d <= ndst; ++d) cnt[d] += cnt[d - 1];
        red_srcs.resize((size_t)cnt[ndst]);
      }
      for (int c = 0; c < h->n_dense; ++c) {
        const ChunkHdr& H = chunks[(size_t)c];
        const int32_t* cams = chunk_cams.data() + H.cam0;
        const int nb = H.ncam * (H.ncam + 1) / 2;
        for (int cj = 0; cj < H.ncam; ++cj) {
          for (int ci = 0; ci <= cj; ++ci) {
            const int64_t b = host_sky(cams[ci], cams[cj]);
            if (b < 0) continue;  // two cameras of the chunk that share no landmark anywhere: their product is exactly zero, S has no such block
            if (b >= nsb) return fail(MPSFM_EUNSUPPORTED, "internal: block index beyond S");
            if (pass == 0) cnt[(size_t)b + 1]++;
            else red_srcs[(size_t)cnt[(size_t)b]++] = H.slab0 + 2 * (cj * (cj + 1) / 2 + ci);
          }
          const size_t d = (size_t)nsb + (size_t)cams[cj];
          if (pass == 0) cnt[d + 1]++;
          else red_srcs[(size_t)cnt[d]++] = H.slab0 + 2 * nb + cj;
        }
      }
    }
    // after the placing pass cnt[d] is the END of destination d
    std::vector<uint8_t> is_diag((size_t)nsb, 0);
    for (int sl = 0; sl < h->ncv; ++sl) {
      if (use_graph && h->plan.nat_of_slot[(size_t)sl] < 0) continue;
      const int64_t b = host_sky(sl, sl);
      if (b >= 0 && b < nsb) is_diag[(size_t)b] = 1;
    }
    constexpr int kPart = 16;
    for (size_t d = 0; d < ndst && !slab_tables_on_device; ++d) {
      const int32_t s0 = d == 0 ? 0 : cnt[d - 1], s1 = cnt[d];
      for (int32_t q = s0; q < s1; q += kPart) {
        RedDest R;
        R.kind = d >= (size_t)nsb ? 2 : (is_diag[d] ? 1 : 0);
        R.dst = d >= (size_t)nsb ? (int32_t)(d - (size_t)nsb) : (int32_t)d;
        R.s0 = q; R.s1 = std::min(q + kPart, s1);
        red_dests.push_back(R);
      }
    }
    h->n_red_dests = (int)red_dests.size();
    h->n_red_srcs = (int64_t)red_srcs.size(); h->n_chunk_cams = (int64_t)chunk_cams.size();
  }
  lap("slab reduction tables");

  // -- upload
  int rc = 0;
  std::vector<double> intr(P->cam_intr, P->cam_intr + (size_t)P->n_intr * 4);
  std::vector<int32_t> intr_idx(P->cam_intr_idx, P->cam_intr_idx + nc);
  if ((rc = dev_upload(&h->d_intr, intr))) return rc;
  if ((rc = dev_upload(&h->d_intr_idx, intr_idx))) return rc;
  if ((rc = dev_upload(&h->d_cmask, cmask))) return rc;
  if ((rc = dev_upload(&h->d_cam_slot, h->cam_slot_h))) return rc;
  {
    std::vector<int32_t> cam_of_slot((size_t)std::max(h->ncv, 1), 0);
    for (int i = 0; i < nc; ++i) if (h->cam_slot_h[(size_t)i] >= 0 && h->cam_slot_h[(size_t)i] < h->ncv) cam_of_slot[(size_t)h->cam_slot_h[(size_t)i]] = i;
    if ((rc = dev_upload(&h->d_cam_of_slot, cam_of_slot))) return rc;
  }
  if ((rc = dev_upload(&h->d_chunks, chunks))) return rc;
  if ((rc = dev_upload(&h->d_chunk_cams, chunk_cams))) return rc;
  if (h->np > 0 && h->np == (int64_t)h->np_user) {
    if ((rc = dev_upload(&h->d_perm, h->perm))) return rc;
    if ((rc = dev_alloc(&h->d_user_pts, (size_t)h->np * 3))) return rc;
  }
  h->built_on_device = dev;
  if (dev) {  // the device build's tables are where they belong
    h->d_rec_cam = DB.d_rec_cam; h->d_rec_pt = DB.d_rec_pt; h->d_rec_meta = DB.d_rec_meta; h->d_rec_xy = DB.d_rec_xy; h->d_rec_d = DB.d_rec_d;
    h->d_rec_m = DB.d_rec_m; h->d_rec_a = DB.d_rec_a; h->d_pt_rec_start = DB.d_pt_rec_start; h->d_pt_kv = DB.d_pt_kv;
    h->d_fx_cam = DB.d_fx_cam; h->d_fx_pt = DB.d_fx_pt; h->d_fx_meta = DB.d_fx_meta; h->d_fx_xy = DB.d_fx_xy; h->d_fx_d = DB.d_fx_d; h->d_fx_m = DB.d_fx_m;
    h->d_fx_a = DB.d_fx_a;
    DB.d_rec_cam = DB.d_rec_pt = DB.d_pt_rec_start = DB.d_fx_cam = DB.d_fx_pt = nullptr; DB.d_rec_meta = DB.d_fx_meta = nullptr; DB.d_pt_kv = nullptr;
    DB.d_rec_xy = DB.d_rec_d = DB.d_rec_m = DB.d_rec_a = DB.d_fx_xy = DB.d_fx_d = DB.d_fx_m = DB.d_fx_a = nullptr;
    DB.release();  // the device copies of chunks / camera lists: the host copies (slab offsets added) are uploaded above
  } else {
  if ((rc = dev_upload(&h->d_rec_cam, rec_cam))) return rc;
  if ((rc = dev_upload(&h->d_rec_pt, rec_pt))) return rc;
  if ((rc = dev_upload(&h->d_rec_meta, rec_meta))) return rc;
  if ((rc = dev_upload(&h->d_rec_xy, rec_xy))) return rc;
  if ((rc = dev_upload(&h->d_rec_d, rec_d))) return rc;
  if ((rc = dev_upload(&h->d_rec_m, rec_m))) return rc;
  if ((rc = dev_upload(&h->d_rec_a, rec_a))) return rc;
  if ((rc = dev_upload(&h->d_pt_rec_start, pt_rec_start))) return rc;
  if ((rc = dev_upload(&h->d_pt_kv, pt_kv))) return rc;
  if ((rc = dev_upload(&h->d_fx_cam, fx_cam))) return rc;
  if ((rc = dev_upload(&h->d_fx_pt, fx_pt))) return rc;
  if ((rc = dev_upload(&h->d_fx_meta, fx_meta))) return rc;
  if ((rc = dev_upload(&h->d_fx_xy, fx_xy))) return rc;
  if ((rc = dev_upload(&h->d_fx_d, fx_d))) return rc;
  if ((rc = dev_upload(&h->d_fx_m, fx_m))) return rc;
  if ((rc = dev_upload(&h->d_fx_a, fx_a))) return rc;
  }
  if ((rc = dev_upload(&h->d_lhdr, lhdr))) return rc;
  if ((rc = dev_alloc(&h->d_wl, (size_t)std::max<int64_t>(wl_rows, 1) * 18))) return rc;
  h->n_blk_desc = (int64_t)blk_desc.size(); h->n_blk_ent_start = (int64_t)blk_ent_start.size(); h->n_ents = (int64_t)ents.size();
  if ((rc = dev_upload(&h->d_blk_desc, blk_desc))) return rc;
  if ((rc = dev_upload(&h->d_blk_ent_start, blk_ent_start))) return rc;
  if ((rc = dev_upload(&h->d_ents, ents))) return rc;
  int32_t* d_diag_block = nullptr;
  if (!slab_tables_on_device) {
    if ((rc = dev_upload(&h->d_red_dests, red_dests))) return rc;
    if ((rc = dev_upload(&h->d_red_srcs, red_srcs))) return rc;
  } else if ((rc = dev_upload(&d_diag_block, diag_block))) return rc;
  if ((rc = dev_alloc(&h->d_slab, (size_t)std::max<int64_t>(slab_units, 1) * 18))) return rc;

  if ((rc = drain_uploads())) { cached_free(d_diag_block); return rc; }
  if (slab_tables_on_device) {
    const BlockSky sky{h->d_sky_first, h->d_sky_start, h->d_sky_index, h->ncv};
    int32_t nd = 0; int64_t ns = 0;
    rc = devb->slab_tables(h->d_chunks, h->n_dense, h->d_chunk_cams, sky, h->sblk_blocks, h->ncv, d_diag_block, &h->d_red_dests, &nd, &h->d_red_srcs, &ns);
    HIP_TRY(hipStreamSynchronize(h->stream));  // d_diag_block goes back to the process-wide cache
    cached_free(d_diag_block);
    if (rc) return rc;
    h->n_red_dests = nd; h->n_red_srcs = ns;
  }
  lap("upload tables");
  const size_t ncs = (size_t)std::max(nc, 1), nps = (size_t)std::max<int64_t>(h->np, 1);
  for (double** p : {&h->d_q, &h->d_q2, &h->d_q0}) if ((rc = dev_alloc(p, ncs * 4))) return rc;
  for (double** p : {&h->d_t, &h->d_t2, &h->d_t0}) if ((rc = dev_alloc(p, ncs * 3))) return rc;
  for (double** p : {&h->d_pts, &h->d_pts2, &h->d_pts0, &h->d_ps, &h->d_diagV}) if ((rc = dev_alloc(p, nps * 3))) return rc;
  if ((rc = dev_alloc(&h->d_cs, ncs * 6))) return rc;
  if ((rc = dev_alloc(&h->d_camtab, ncs * kCamRec))) return rc;
  if ((rc = dev_alloc(&h->d_camtab2, ncs * kCamRec))) return rc;
  h->sblk_count = h->sblk_blocks * 36;
  h->red_count = h->sblk_count + 3 * (int64_t)h->n_user + SC_COUNT;
  if ((rc = dev_alloc(&h->d_red, (size_t)h->red_count))) return rc;
  h->d_Sblk = h->d_red; h->d_gc = h->d_red + h->sblk_count; h->d_wv = h->d_gc + h->n_user; h->d_diagU = h->d_wv + h->n_user;
  h->d_redsc = h->d_diagU + h->n_user;
  if ((rc = dev_alloc(&h->d_part, (size_t)std::max(h->nchunks + h->nlong, 1) * 4))) return rc;
  if ((rc = dev_alloc(&h->d_part2, (size_t)std::max(h->nchunks + h->nlong, 1) * 8))) return rc;
  if ((rc = dev_alloc(&h->d_scal, (size_t)U_COUNT))) return rc;
  if ((rc = dev_alloc(&h->d_costpart, (size_t)1024 * 4))) return rc;
  static_assert(sizeof(double) * U_COUNT * 2 <= HandleResources::kPinnedBytes, "pinned scalar block too small");
  HIP_TRY(pooled_pinned((void**)&h->h_scal));
  static_assert(sizeof(LmCtl) * 2 <= HandleResources::kPinnedBytes, "pinned block too small for two control-block copies");
  HIP_TRY(pooled_pinned((void**)&h->h_ctl));
  if ((rc = dev_alloc(&h->d_ctl, 1))) return rc;
  for (auto& e : h->ev2) HIP_TRY(pooled_event(&e, true));
  const size_t ntiles = (size_t)(h->nt + 1) * (h->nt + 2) / 2;
  if ((rc = dev_alloc(&h->d_A, ntiles * 1024))) return rc;
  if ((rc = dev_alloc(&h->d_dwork, dense_work_doubles(h->nt)))) return rc;
  if ((rc = dev_alloc(&h->d_yc, (size_t)std::max(h->n_user, 1)))) return rc;
  if ((rc = dev_alloc(&h->d_fail, 1))) return rc;
  HIP_TRY(hipMemsetAsync(h->d_fail, 0, sizeof(int), h->stream));
  for (auto& e : h->ev) HIP_TRY(pooled_event(&e, true));
  // tuning / test overrides of the dense factorisation, read once per handle
  if (const char* e = std::getenv("MPSFM_CHOL_NB")) h->ov.nb = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("MPSFM_CHOL_BIG")) h->ov.big = std::atoi(e) != 0;
  if (const char* e = std::getenv("MPSFM_CHOL_OVERLAP")) h->ov.overlap = std::atoi(e) != 0;
  if (const char* e = std::getenv("MPSFM_CHOL_INVERSE")) h->ov.no_inverse = std::atoi(e) == 0;
  if (const char* e = std::getenv("MPSFM_CHOL_LEVEL")) h->ov.no_level = std::atoi(e) == 0;
  // a large reduced system without exploitable structure (every camera shares landmarks with most others): the
  // outer-panel path with its LDS-staged 64x64 trailing update moves fewer bytes per flop than one workgroup per tile
  else if (h->nt > dense_plain_max_tiles() && (double)h->plan.products > 0.5 * (double)h->nt * h->nt * h->nt / 6.0) h->ov.no_level = true;
  if (h->nt > 64 || h->ov.nb > 0) {
    HIP_TRY(pooled_stream(&h->ov.s2));
    for (auto& e : h->ov.evF) HIP_TRY(pooled_event(&e, false));
    for (auto& e : h->ov.evB) HIP_TRY(pooled_event(&e, false));
  }
  // Small problems (local bundle adjustment): the whole trust-region loop in one cooperative launch, one workgroup per chunk
  h->local_ok = false;
  {
    const char* e = std::getenv("MPSFM_LOCAL_LM");
    const bool wanted = !(e && std::atoi(e) == 0);
    if (wanted && !sharded(h) && h->nlong == 0 && h->nchunks > 0 && h->n_dense == h->nchunks && h->ncv >= 1 && h->ncv <= kLocalCams &&
        h->n_user == 6 * h->ncv && h->nchunks <= local_lm_max_chunks(h->device)) {
      h->local_log_cap = std::max(h->opt.max_num_iterations, 0) + 2;
      if ((rc = dev_alloc(&h->d_local_acc, (size_t)2 * kLocalAccDoubles))) return rc;
      if ((rc = dev_alloc(&h->d_local_sync, (size_t)16))) return rc;
      if ((rc = dev_alloc(&h->d_local_log, (size_t)h->local_log_cap))) return rc;
      h->local_ok = true;
    }
  }
  HIP_TRY(hipMemsetAsync(h->d_ps, 0, nps * 3 * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->d_yc, 0, (size_t)std::max(h->n_user, 1) * sizeof(double), h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  init_tile_tables(h->stream);
  (void)st;
  lap("allocate work buffers");
  return 0;
}

static int upload_state(mpsfm_ba_handle* h, const mpsfm_ba_state* st, bool as_initial) {
  if (!st || (h->nc > 0 && (!st->cam_quat_xyzw || !st->cam_t)) || (h->np > 0 && !st->pts)) return fail(MPSFM_EINVAL, "state is NULL");
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (h->opt.verbose < 2) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mpsfm_ba] state: %-28s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  // caller memory is pageable: staged copies (see Stager).  The handle's stream is idle between solves.
  HIP_TRY(hipStreamSynchronize(h->stream));
  lap("stream idle");
  if (h->nc > 0) {
    if (int rc = staged_h2d(h->d_q, st->cam_quat_xyzw, sizeof(double) * 4 * h->nc)) return rc;
    if (int rc = staged_h2d(h->d_t, st->cam_t, sizeof(double) * 3 * h->nc)) return rc;
  }
  lap("pose copies");
  if (h->d_perm) {  // every landmark is referenced: the caller's array as it is, re-ordered on the device
    if (int rc = staged_h2d(h->d_user_pts, st->pts, sizeof(double) * 3 * h->np)) return rc;
    launch_permute_pts(h->np, h->d_perm, h->d_user_pts, h->d_pts, false, h->stream);
    lap("landmark copy + permute (device)");
  } else {
    std::vector<double> sorted((size_t)h->np * 3);
    parallel_ranges(h->np, 16384, [&](int64_t k0, int64_t k1) {
      for (int64_t k = k0; k < k1; ++k) {
        const double* s = st->pts + 3 * (size_t)h->perm[(size_t)k];
        sorted[3 * (size_t)k] = s[0]; sorted[3 * (size_t)k + 1] = s[1]; sorted[3 * (size_t)k + 2] = s[2];
      }
    });
    lap("permute landmarks");
    if (h->np > 0) if (int rc = staged_h2d(h->d_pts, sorted.data(), sizeof(double) * 3 * h->np)) return rc;
  }
  lap("landmark copy");
  if (as_initial) {
    HIP_TRY(hipMemcpyAsync(h->d_q0, h->d_q, sizeof(double) * 4 * h->nc, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_t0, h->d_t, sizeof(double) * 3 * h->nc, hipMemcpyDeviceToDevice, h->stream));
    if (h->np > 0) HIP_TRY(hipMemcpyAsync(h->d_pts0, h->d_pts, sizeof(double) * 3 * h->np, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    lap("keep initial state");
  }
  h->scales_ready = false;
  return 0;
}

static SweepArgs sweep_args(mpsfm_ba_handle* h, double radius, const LmCtl* ctl = nullptr) {
  SweepArgs a{};
  a.ctl = ctl;
  a.chunks = h->d_chunks; a.chunk_cams = h->d_chunk_cams; a.rec_cam = h->d_rec_cam; a.rec_meta = h->d_rec_meta;
  a.rec_xy = h->d_rec_xy; a.rec_d = h->d_rec_d; a.rec_m = h->d_rec_m; a.rec_a = h->d_rec_a;
  a.pt_rec_start = h->d_pt_rec_start; a.pt_kv = h->d_pt_kv; a.blk_desc = h->d_blk_desc; a.blk_ent_start = h->d_blk_ent_start; a.ents = h->d_ents;
  a.camtab = h->d_camtab; a.pts = h->d_pts; a.ps = h->d_ps; a.loss = h->loss;
  a.radius = radius; a.min_diag = h->opt.min_lm_diagonal; a.max_diag = h->opt.max_lm_diagonal; a.ncv = h->ncv; a.dbg = (g_dbg_flags >> 8) & 0xff;
  a.lhdr = h->d_lhdr; a.nlong = h->nlong; a.nchunks = h->nchunks; a.cam_slot = h->d_cam_slot; a.wl = h->d_wl;
  a.sky = BlockSky{h->d_sky_first, h->d_sky_start, h->d_sky_index, h->ncv};
  a.Sblk = h->d_Sblk; a.gc = h->d_gc; a.wv = h->d_wv; a.diagU = h->d_diagU; a.part = h->d_part; a.diagV = h->d_diagV;
  a.yc = h->d_yc; a.camtab2 = h->d_camtab2; a.pts2 = h->d_pts2; a.part2 = h->d_part2;
  a.slab = h->d_slab; a.chunk0 = 0;
  return a;
}

// cost of a record list; out[0] reprojection, out[1] depth, out[2] bad count (host values)
static int cost_of_records(mpsfm_ba_handle* h, int64_t nrec, const int32_t* cam, const int32_t* pt, const uint32_t* meta,
                           const double* xy, const double* d, const double* m, const double* a, double* out3) {
  out3[0] = out3[1] = out3[2] = 0.0;
  if (nrec <= 0) return 0;
  const int nb = (int)std::min<int64_t>(1024, (nrec + kThreads - 1) / kThreads);
  CostArgs c{nrec, cam, pt, meta, xy, d, m, a, h->d_camtab, h->d_pts, h->loss, h->d_costpart};
  launch_cost_records(c, nb, h->stream);
  launch_reduce_cols(h->d_costpart, nb, 4, 3, 0u, h->d_scal, h->stream);
  HIP_TRY(hipMemcpyAsync(h->h_scal, h->d_scal, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  out3[0] = h->h_scal[0]; out3[1] = h->h_scal[1]; out3[2] = h->h_scal[2];
  return 0;
}

// Jacobi column scales from the Jacobian at the current state (Ceres: iteration 0 only)
static int prepare_scales(mpsfm_ba_handle* h) {
  hipStream_t s = h->stream;
  launch_cam_scales(h->nc, h->d_cam_slot, h->d_cmask, h->d_diagU, 0, h->d_cs, s);
  launch_pt_scales(h->np, h->d_pt_kv, h->d_diagV, 0, h->d_ps, s);
  launch_build_camtab(h->nc, h->d_q, h->d_t, h->d_intr, h->d_intr_idx, h->d_cs, h->d_camtab, s);
  if (h->opt.jacobi_scaling) {
    HIP_TRY(hipMemsetAsync(h->d_diagU, 0, sizeof(double) * (size_t)std::max(h->n_user, 1), s));
    HIP_TRY(hipMemsetAsync(h->d_diagV, 0, sizeof(double) * 3 * (size_t)std::max<int64_t>(h->np, 1), s));
    SweepArgs a = sweep_args(h, 1.0);
    launch_track_sweep(a, h->nchunks, true, s);
    if (int rc = allreduce_dev(h, h->d_diagU, h->n_user)) return rc;
    launch_cam_scales(h->nc, h->d_cam_slot, h->d_cmask, h->d_diagU, 1, h->d_cs, s);
    launch_pt_scales(h->np, h->d_pt_kv, h->d_diagV, 1, h->d_ps, s);
    launch_build_camtab(h->nc, h->d_q, h->d_t, h->d_intr, h->d_intr_idx, h->d_cs, h->d_camtab, s);
  }
  HIP_TRY(hipGetLastError());
  h->scales_ready = true;
  return 0;
}

// the full track sweep: dense chunks through their slabs (k_track_sweep_dense + k_reduce_slabs), the others and the long
// tracks through the general kernels (global atomics); all of them add into the zeroed reduced buffer
static void launch_sweeps(mpsfm_ba_handle* h, SweepArgs a, hipStream_t s) {
  launch_track_sweep_dense(a, h->n_dense, s);
  launch_reduce_slabs(h->d_red_dests, h->n_red_dests, h->d_red_srcs, h->d_slab, h->d_Sblk, h->d_gc, h->d_wv, h->d_diagU, a.ctl, s);
  a.chunk0 = h->n_dense;
  launch_track_sweep(a, h->nchunks - h->n_dense, false, s);
}

// one track sweep at the current state: fills the reduced buffer and its scalar tail
// (inside the solve loop the prologue kernel has zeroed the buffer; single-rank runs reduce the partials with the decision)
static int run_track_sweep(mpsfm_ba_handle* h, double radius, const LmCtl* ctl = nullptr, bool in_loop = false, bool adopt = false) {
  hipStream_t s = h->stream;
  if (!in_loop) launch_zero(h->d_red, h->red_count, ctl, s);
  SweepArgs a = sweep_args(h, radius, ctl);
  if (adopt) {
    a.adopt_on = 1; a.adopt_nc = h->nc;
    a.pts_rw = h->d_pts; a.q_rw = h->d_q; a.t_rw = h->d_t; a.camtab_rw = h->d_camtab; a.q2 = h->d_q2; a.t2 = h->d_t2;
    a.red = h->d_red; a.nred = h->red_count;
  }
  launch_sweeps(h, a, s);
  if (h->nchunks + h->nlong > 0 && !(in_loop && !sharded(h)))
    launch_reduce_cols(h->d_part, h->nchunks + h->nlong, 4, 3, 1u << 2, h->d_redsc, s, sharded(h) ? nullptr : h->d_scal + U_X_COST,
                       sharded(h) ? (h->opt.rank > 0 ? h->opt.rank : 0) % kMaxRankSlots : -1);
  else if (sharded(h)) launch_gmax_to_slot(h->d_redsc, h->opt.rank > 0 ? h->opt.rank : 0, s);  // (a rank without chunks: its slot from the zeroed buffer)
  h->last_radius = radius;
  return 0;
}

static int run_dense(mpsfm_ba_handle* h, double radius, const LmCtl* ctl = nullptr) {
  hipStream_t s = h->stream;
  // d_fail is zero here: cleared at creation and re-armed by k_cam_update after every read
  if (h->n > 0) {
    // the level-scheduled factorisation without inverse accumulators only touches the tiles of its plan
    const bool level = dense_level(&h->ov, &h->lp);
    double* pinv = dense_pinv(h->d_dwork, h->nt, &h->ov, &h->lp);
    const bool listed = level && !pinv;
    AssembleArgs as{BlockSky{h->d_sky_first, h->d_sky_start, h->d_sky_index, h->ncv}, h->d_Sblk, h->d_gc, h->d_wv, h->d_diagU, h->ncv, h->n, h->nt, radius,
                    h->opt.min_lm_diagonal, h->opt.max_lm_diagonal, h->d_A, pinv, listed ? h->lp.d_asm_tiles : nullptr, listed ? h->lp.n_asm : 0,
                    h->lp.d_col_slot, h->n_user, (level && pinv) ? h->d_yc : nullptr, (level && pinv) ? h->lp.d_tile_live : nullptr, ctl};
    launch_assemble(as, s);
    launch_dense_solve(h->d_A, h->d_dwork, h->nt, h->n, h->d_yc, h->d_fail, s, &h->ov, &h->lp, ctl);
  }
  return 0;
}

// Small problems (local bundle adjustment): fixed cost, column scales, the trust-region loop and the state norm without a single
// host synchronisation before the end — the loop is ONE cooperative launch (local_lm.hip).  Returns kLocalRefused when the launch
// is not accepted (nothing has changed the state then: the launch chain takes over).
constexpr int kLocalRefused = 1;
static int solve_local(mpsfm_ba_handle* h, mpsfm_ba_summary* sum) {
  hipStream_t s = h->stream;
  const mpsfm_ba_options& o = h->opt;
  launch_cam_scales(h->nc, h->d_cam_slot, h->d_cmask, h->d_diagU, 0, h->d_cs, s);
  launch_build_camtab(h->nc, h->d_q, h->d_t, h->d_intr, h->d_intr_idx, h->d_cs, h->d_camtab, s);
  double* fixed_parts = h->d_scal + 12;  // three free slots of the scalar block
  HIP_TRY(hipMemsetAsync(h->d_scal, 0, sizeof(double) * U_COUNT, s));
  if (h->nfixed > 0) {
    const int nb = (int)std::min<int64_t>(1024, (h->nfixed + kThreads - 1) / kThreads);
    CostArgs c{h->nfixed, h->d_fx_cam, h->d_fx_pt, h->d_fx_meta, h->d_fx_xy, h->d_fx_d, h->d_fx_m, h->d_fx_a, h->d_camtab, h->d_pts, h->loss, h->d_costpart};
    launch_cost_records(c, nb, s);
    launch_reduce_cols(h->d_costpart, nb, 4, 3, 0u, fixed_parts, s);
  }
  if (int rc = prepare_scales(h)) return rc;
  LmCtl c0;
  std::memset(&c0, 0, sizeof(c0));
  c0.radius = o.initial_trust_region_radius; c0.decrease_factor = 2.0;
  c0.term = kLmRunning; c0.check_gradient = 1;  // x_norm and fixed_cost: filled in by the launch
  h->h_ctl[0] = c0;
  HIP_TRY(hipMemcpyAsync(h->d_ctl, &h->h_ctl[0], sizeof(LmCtl), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(h->d_local_acc, 0, sizeof(double) * 2 * kLocalAccDoubles, s));
  HIP_TRY(hipMemsetAsync(h->d_local_sync, 0, sizeof(int64_t) * 16, s));
  LocalArgs la{};
  la.A = sweep_args(h, 0.0, nullptr);
  la.ctl = h->d_ctl;
  la.o = LmOpts{o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance, o.min_relative_decrease, o.max_trust_region_radius,
                o.min_trust_region_radius, o.max_num_iterations, o.max_num_consecutive_invalid_steps};
  la.log = o.verbose > 0 ? h->d_local_log : nullptr;
  la.acc[0] = h->d_local_acc; la.acc[1] = h->d_local_acc + kLocalAccDoubles;
  la.bar = reinterpret_cast<int32_t*>(h->d_local_sync); la.clk = reinterpret_cast<long long*>(h->d_local_sync + 1);
  la.ncv = h->ncv; la.nc = h->nc; la.nchunks = h->nchunks;
  la.q = h->d_q; la.t = h->d_t; la.camtab = h->d_camtab; la.pts = h->d_pts; la.cs = h->d_cs; la.fixed_parts = fixed_parts;
  if (launch_local_lm(la, s) != (int)hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(hipStreamSynchronize(s));  // the pinned control block is free again
    return kLocalRefused;
  }
  static_assert(sizeof(int64_t) * 8 <= sizeof(double) * U_COUNT * 2, "the pinned scalar block holds the sync words");
  int64_t* hs = reinterpret_cast<int64_t*>(h->h_scal);
  HIP_TRY(hipMemcpyAsync(&h->h_ctl[0], h->d_ctl, sizeof(LmCtl), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hs, h->d_local_sync, sizeof(int64_t) * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (reinterpret_cast<const int32_t*>(hs)[1] != 0)
    return fail(MPSFM_EHIP, "single-launch solver: a grid barrier did not complete (workgroups not co-resident?); MPSFM_LOCAL_LM=0 selects the launch chain");
  const LmCtl& last = h->h_ctl[0];
  const double fixed = last.fixed_cost;
  sum->fixed_cost = fixed;
  sum->time_linearize_s = 1e-8 * (double)(hs[1] + hs[2]); sum->time_dense_s = 1e-8 * (double)hs[3]; sum->time_update_s = 1e-8 * (double)(hs[4] + hs[5] + hs[6]);
  if (o.verbose > 0) {
    const int nlog = std::min(last.iter + 1, h->local_log_cap);
    std::vector<LmHead> log((size_t)std::max(nlog, 0));
    if (nlog > 0) HIP_TRY(hipMemcpy(log.data(), h->d_local_log, sizeof(LmHead) * (size_t)nlog, hipMemcpyDeviceToHost));
    for (int i = 0; i < nlog; ++i) {
      const LmHead& l = log[(size_t)i];
      if (l.iter != i + 1 && i != nlog - 1) continue;
      if (l.last_mcc > 0.0 && l.last_cand != DBL_MAX)
        std::fprintf(stderr, "[mpsfm_ba] it %3d cost %.9e cand %.9e rel %.3e radius %.3e |step| %.3e\n", i + 1, l.last_x_cost + fixed, l.last_cand + fixed,
                     l.last_rel, l.radius, l.last_step_norm);
      else
        std::fprintf(stderr, "[mpsfm_ba] it %3d invalid step (chol_fail=%d mcc=%.3e) radius %.3e\n", i + 1, l.last_chol_fail, l.last_mcc, l.radius);
    }
  }
  h->last_radius = last.radius;
  sum->num_jacobian_evals = last.n_jac_evals;
  sum->num_residual_evals = (int64_t)h->nblocks_reduced_global * ((int64_t)last.n_cost_evals + last.n_jac_evals);
  if (last.term == kLmNumericError)
    return fail(MPSFM_ENUMERIC, "the initial point cannot be evaluated (non-finite residual or depth <= 0 in a log-depth block)");
  sum->initial_cost = last.initial_cost;
  sum->fixed_cost = last.fixed_cost;  // (one rank: formed on the device)
  sum->final_cost = last.cur_cost + last.fixed_cost;
  sum->num_iterations = last.iter;
  sum->num_successful_steps = last.n_success;
  sum->num_unsuccessful_steps = last.n_unsuccess;
  sum->termination = last.term;
  sum->final_radius = last.radius;
  sum->trace_len = last.trace_len;
  for (int i = 0; i < last.trace_len; ++i) { sum->trace_cost[i] = last.trace_cost[i]; sum->trace_radius[i] = last.trace_radius[i]; sum->trace_accepted[i] = last.trace_accepted[i]; }
  return 0;
}

static int solve_impl(mpsfm_ba_handle* h, mpsfm_ba_summary* sum) {
  using clk = std::chrono::steady_clock;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const mpsfm_ba_options& o = h->opt;
  std::memset(sum, 0, sizeof(*sum));
  HIP_TRY(hipStreamSynchronize(s));
  const auto t_begin = clk::now();
  sum->num_residual_blocks = (int64_t)h->nblocks_global;
  sum->reduced_dim = h->n_user;
  int64_t n_cost_evals = 0, n_jac_evals = 0;
  if (h->local_ok) {
    const int rc = solve_local(h, sum);
    if (rc != kLocalRefused) {
      sum->time_total_s = std::chrono::duration<double>(clk::now() - t_begin).count();
      return rc;
    }
  }

  auto finish = [&](int rc) {
    sum->num_jacobian_evals = n_jac_evals;
    sum->num_residual_evals = (int64_t)h->nblocks_reduced_global * (n_cost_evals + n_jac_evals);
    sum->time_total_s = std::chrono::duration<double>(clk::now() - t_begin).count();
    return rc;
  };
  // Nothing in front of the loop needs the host — the cost of the fixed blocks and the state norm stay on the device and enter
  // the control block there (k_lm_init); three stream synchronisations less per solve.  Sharded runs sum them over the ranks
  // in one small exchange on the device.
  const bool nothing_to_solve = h->n == 0 && h->nvarpts_global == 0.0;
  const bool async_pre = !nothing_to_solve;
  double fixed = 0.0, x_norm = 0.0;
  // camera table at the initial point (unit scales) for the fixed cost
  launch_cam_scales(h->nc, h->d_cam_slot, h->d_cmask, h->d_diagU, 0, h->d_cs, s);
  launch_build_camtab(h->nc, h->d_q, h->d_t, h->d_intr, h->d_intr_idx, h->d_cs, h->d_camtab, s);
  if (async_pre) {
    HIP_TRY(hipMemsetAsync(h->d_scal, 0, sizeof(double) * U_COUNT, s));
    if (h->nfixed > 0) {
      const int nb = (int)std::min<int64_t>(1024, (h->nfixed + kThreads - 1) / kThreads);
      CostArgs c{h->nfixed, h->d_fx_cam, h->d_fx_pt, h->d_fx_meta, h->d_fx_xy, h->d_fx_d, h->d_fx_m, h->d_fx_a, h->d_camtab, h->d_pts, h->loss, h->d_costpart};
      launch_cost_records(c, nb, s);
      launch_reduce_cols(h->d_costpart, nb, 4, 3, 0u, h->d_scal + 12, s);
    }
  } else {
    double fx[3];
    if (int rc = cost_of_records(h, h->nfixed, h->d_fx_cam, h->d_fx_pt, h->d_fx_meta, h->d_fx_xy, h->d_fx_d, h->d_fx_m, h->d_fx_a, fx)) return rc;
    fixed = fx[0] + fx[1];
    if (int rc = allreduce_host(h, &fixed, 1)) return rc;
    sum->fixed_cost = fixed;
  }

  if (nothing_to_solve) {
    double c3[3];
    if (int rc = cost_of_records(h, h->nrec, h->d_rec_cam, h->d_rec_pt, h->d_rec_meta, h->d_rec_xy, h->d_rec_d, h->d_rec_m, h->d_rec_a, c3)) return rc;
    double c = c3[0] + c3[1];
    if (int rc = allreduce_host(h, &c, 1)) return rc;
    sum->initial_cost = sum->final_cost = c + fixed;
    sum->termination = MPSFM_TERM_NO_VARIABLES;
    return finish(0);
  }

  if (int rc = prepare_scales(h)) return rc;
  if (h->np > 0) HIP_TRY(hipMemcpyAsync(h->d_pts2, h->d_pts, sizeof(double) * 3 * h->np, hipMemcpyDeviceToDevice, s));
  // initial x norm: cameras through a zero-step camera update, landmarks by a reduction
  HIP_TRY(hipMemsetAsync(h->d_yc, 0, sizeof(double) * (size_t)std::max(h->n_user, 1), s));
  if (!async_pre) HIP_TRY(hipMemsetAsync(h->d_scal, 0, sizeof(double) * U_COUNT, s));
  {
    HIP_TRY(hipMemsetAsync(h->d_gc, 0, sizeof(double) * (size_t)std::max(h->n_user, 1), s));
    launch_cam_update(h->nc, h->d_cam_slot, h->d_q, h->d_t, h->d_cs, h->d_yc, h->d_gc, h->d_q2, h->d_t2, h->d_scal, s);
    const int nb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (h->np + kThreads - 1) / kThreads));
    launch_pts_sqnorm(h->np, h->d_pt_kv, h->d_pts, h->d_costpart, nb, s);
    launch_reduce_cols(h->d_costpart, nb, 1, 1, 0u, h->d_scal + U_XN_SQ_PTS, s);
    if (!async_pre) {
      HIP_TRY(hipMemcpyAsync(h->h_scal, h->d_scal, sizeof(double) * U_COUNT, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      double v = h->h_scal[U_XN_SQ_PTS];
      if (int rc = allreduce_host(h, &v, 1)) return rc;
      x_norm = std::sqrt(v + h->h_scal[U_XN_SQ_CAMS]);
    }
  }

  // ---- Levenberg-Marquardt loop.  The decisions are taken on the device (k_lm_decide, LmCtl in common.h): the host
  // enqueues iteration i+1 BEFORE it looks at the outcome of iteration i, so the stream never runs dry, and only reads a
  // pinned copy of the control block one iteration late.  When that copy says the solve is over, the one iteration queued
  // ahead returns at once in every kernel.  Every rank of a sharded run sees the same decisions at the same iteration, so
  // all of them enqueue the same sequence of collectives.
  {
    LmCtl c0;
    std::memset(&c0, 0, sizeof(c0));
    c0.radius = o.initial_trust_region_radius; c0.decrease_factor = 2.0; c0.x_norm = x_norm; c0.fixed_cost = fixed;
    c0.term = kLmRunning; c0.check_gradient = 1;
    h->h_ctl[0] = c0;
    HIP_TRY(hipMemcpyAsync(h->d_ctl, &h->h_ctl[0], sizeof(LmCtl), hipMemcpyHostToDevice, s));
    if (async_pre) {  // x_norm and fixed_cost from the device scalars
      double* sums = nullptr;
      if (sharded(h)) {
        sums = h->d_costpart;  // (free again: its reductions are queued in front)
        launch_lm_pack(h->d_scal, sums, s);
        if (int rc = allreduce_dev(h, sums, 3)) return rc;
      }
      launch_lm_init(h->d_ctl, h->d_scal, sums, s);
    }
    else HIP_TRY(hipStreamSynchronize(s));  // the pinned slot is reused below
  }
  const LmOpts lo{o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance, o.min_relative_decrease, o.max_trust_region_radius,
                  o.min_trust_region_radius, o.max_num_iterations, o.max_num_consecutive_invalid_steps};
  const LmCtl* ctl = h->d_ctl;
  const double host_radius = 0.0;  // unused: the kernels read the radius from the control block
  const bool fuse_prologue = [] { const char* e = std::getenv("MPSFM_FUSE_PROLOGUE"); return !(e && std::atoi(e) == 0); }();
  const bool fuse_cam = [] { const char* e = std::getenv("MPSFM_FUSE_CAM"); return !(e && std::atoi(e) == 0); }();
  auto enqueue_iteration = [&](int it) -> int {
    hipEvent_t* ev = (it & 1) ? h->ev2 : h->ev;
    HIP_TRY(hipEventRecord(ev[0], s));
    // all chunks dense: the dense sweep adopts an accepted candidate and zeroes the reduced buffer itself (sharded runs too: the
    // exchange of the reduced buffer comes behind the slab reduction, the camera scalars are the same on every rank)
    const bool fused_prologue = fuse_prologue && h->nlong == 0 && h->n_dense > 0 && h->n_dense == h->nchunks;
    if (!fused_prologue)
      launch_lm_prologue(h->d_ctl, h->d_red, h->red_count, h->nc, h->np, h->d_q, h->d_t, h->d_camtab, h->d_pts, h->d_q2, h->d_t2, h->d_camtab2, h->d_pts2, s);
    if (int rc = run_track_sweep(h, host_radius, ctl, true, fused_prologue)) return rc;
    if (int rc = allreduce_dev(h, h->d_red, h->red_count)) return rc;
    HIP_TRY(hipEventRecord(ev[1], s));
    if (int rc = run_dense(h, host_radius, ctl)) return rc;
    HIP_TRY(hipEventRecord(ev[2], s));
    // all chunks dense and one rank: the update sweep forms the candidate cameras itself (CamUpdArgs)
    const bool fused_cam = fuse_cam && fused_prologue;
    CamUpdArgs cu{1, h->nc, h->d_cam_slot, h->d_cam_of_slot, h->d_q, h->d_t, h->d_cs, h->d_gc, h->d_intr, h->d_intr_idx,
                  h->d_q2, h->d_t2, h->d_camtab2, h->d_scal, h->d_fail};
    if (!fused_cam)
      launch_cam_update(h->nc, h->d_cam_slot, h->d_q, h->d_t, h->d_cs, h->d_yc, h->d_gc, h->d_q2, h->d_t2, h->d_scal, s,
                        h->d_intr, h->d_intr_idx, h->d_camtab2, h->d_fail, ctl);
    {
      SweepArgs a = sweep_args(h, host_radius, ctl);
      launch_update_sweep(a, h->nchunks, s, fused_cam ? &cu : nullptr);
    }
    if (!sharded(h)) {
      launch_lm_reduce_decide(h->d_part, h->d_part2, h->nchunks + h->nlong, h->d_ctl, h->d_scal, lo, &h->h_ctl[it & 1], s);
    } else {
      if (h->nchunks + h->nlong > 0) launch_reduce_cols(h->d_part2, h->nchunks + h->nlong, 8, 5, 0u, h->d_scal, s);
      if (int rc = allreduce_dev(h, h->d_scal, 5)) return rc;
      // the all-reduced scalars of the track sweep: cost and bad count summed, landmark-gradient maximum over the rank slots
      launch_lm_decide(h->d_ctl, h->d_scal, lo, &h->h_ctl[it & 1], s, h->d_redsc);
    }
    HIP_TRY(hipEventRecord(ev[3], s));
    return 0;
  };
  LmCtl last = h->h_ctl[0];
  if (last.term == kLmRunning) {
    if (int rc = enqueue_iteration(1)) return rc;
    const bool speculate = [] { const char* e = std::getenv("MPSFM_LM_SPECULATE"); return !(e && std::atoi(e) == 0); }();  // per solve: tests switch it
    for (int it = 1;; ++it) {
      if (speculate) { if (int rc = enqueue_iteration(it + 1)) return rc; }  // ahead of the news about iteration `it`
      hipEvent_t* ev = (it & 1) ? h->ev2 : h->ev;
      HIP_TRY(hipEventSynchronize(ev[3]));
      HIP_TRY(hipGetLastError());
      float ms;
      HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); sum->time_linearize_s += 1e-3 * ms;
      HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2])); sum->time_dense_s += 1e-3 * ms;
      HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); sum->time_update_s += 1e-3 * ms;
      const LmCtl prev = last;
      last = h->h_ctl[it & 1];
      if (o.verbose > 0) {
        if (last.last_mcc > 0.0 && last.last_cand != DBL_MAX)
          std::fprintf(stderr, "[mpsfm_ba] it %3d cost %.9e cand %.9e rel %.3e radius %.3e |step| %.3e\n", it, last.last_x_cost + last.fixed_cost,
                       last.last_cand + last.fixed_cost, last.last_rel, last.radius, last.last_step_norm);
        else
          std::fprintf(stderr, "[mpsfm_ba] it %3d invalid step (chol_fail=%d mcc=%.3e) radius %.3e\n", it, last.last_chol_fail, last.last_mcc, last.radius);
      }
      (void)prev;
      if (last.term != kLmRunning) break;
      if (!speculate) { if (int rc = enqueue_iteration(it + 1)) return rc; }
    }
    // the iteration that ended the solve may have been accepted (iteration or radius limit); the copy is idempotent
    launch_lm_accept(h->d_ctl, h->nc, h->np, h->d_q, h->d_t, h->d_camtab, h->d_pts, h->d_q2, h->d_t2, h->d_camtab2, h->d_pts2, s);
    HIP_TRY(hipStreamSynchronize(s));  // the iteration queued ahead has drained (every kernel of it returned at once)
  }
  h->last_radius = last.radius;
  n_cost_evals = last.n_cost_evals; n_jac_evals = last.n_jac_evals;
  if (last.term == kLmNumericError)
    return finish(fail(MPSFM_ENUMERIC, "the initial point cannot be evaluated (non-finite residual or depth <= 0 in a log-depth block)"));
  sum->initial_cost = last.initial_cost;
  sum->fixed_cost = last.fixed_cost;  // (one rank: formed on the device)
  sum->final_cost = last.cur_cost + last.fixed_cost;
  sum->num_iterations = last.iter;
  sum->num_successful_steps = last.n_success;
  sum->num_unsuccessful_steps = last.n_unsuccess;
  sum->termination = last.term;
  sum->final_radius = last.radius;
  sum->trace_len = last.trace_len;
  for (int i = 0; i < last.trace_len; ++i) { sum->trace_cost[i] = last.trace_cost[i]; sum->trace_radius[i] = last.trace_radius[i]; sum->trace_accepted[i] = last.trace_accepted[i]; }
  return finish(0);
}

static thread_local int g_cu_count = 256;
static int create_impl(const mpsfm_ba_problem* P, const mpsfm_ba_state* st, const mpsfm_ba_options* o, mpsfm_ba_handle** out) {
  if (!out) return fail(MPSFM_EINVAL, "out is NULL");
  *out = nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&](const char* what) {
    if (o && o->verbose >= 2)
      std::fprintf(stderr, "[mpsfm_ba] create: %-27s %8.2f ms (cumulative)\n", what,
                   1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  };
  if (int rc = check_problem(P)) return rc;
  if (!o) return fail(MPSFM_EINVAL, "options is NULL");
  since("check_problem");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPSFM_ENODEVICE, "no HIP device visible: libmpsfm_hip has no CPU fallback");
  if (o->device < 0 || o->device >= ndev) return fail(MPSFM_EINVAL, "device ordinal out of range");
  if (o->device >= kMaxDevices) return fail(MPSFM_EUNSUPPORTED, "device ordinals beyond 15 are not supported (per-device pools)");
  {
    // the architecture of a device does not change: query it once per process and device
    static std::mutex mu;
    static std::vector<std::string> arch;
    static std::vector<int> cus;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)arch.size() < ndev) { arch.resize((size_t)ndev); cus.resize((size_t)ndev, 256); }
    if (arch[(size_t)o->device].empty()) {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, o->device));
      arch[(size_t)o->device] = prop.gcnArchName;
      cus[(size_t)o->device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    g_cu_count = cus[(size_t)o->device];
    if (std::strncmp(arch[(size_t)o->device].c_str(), "gfx950", 6) != 0)
      return fail(MPSFM_ENODEVICE, std::string("device is ") + arch[(size_t)o->device] + ", this library is built for gfx950 only");
  }
  since("device check");
  HIP_TRY(hipSetDevice(o->device));
  mpsfm_ba_handle* h = new mpsfm_ba_handle();
  h->device = o->device; h->opt = *o;
  if (o->stream) { h->stream = (hipStream_t)o->stream; h->own_stream = false; }
  else {
    if (pooled_stream(&h->stream) != hipSuccess) { delete h; return fail(MPSFM_EHIP, "hipStreamCreate failed"); }
    h->own_stream = true;
  }
  if (o->use_rccl && o->world_size >= 1) {
    Rccl& R = rccl();
    if (!R.ok) { free_handle(h); return fail(MPSFM_ECOMM, "use_rccl: " + R.why); }
    if (o->rank < 0 || o->rank >= o->world_size) { free_handle(h); return fail(MPSFM_EINVAL, "rank out of range"); }
    Rccl::UniqueId id;
    std::memcpy(id.internal, o->comm_id, sizeof(id.internal));
    const int nrc = R.CommInitRank(&h->comm, o->world_size, id, o->rank);
    if (nrc != 0) { h->comm = nullptr; free_handle(h); return fail(MPSFM_ECOMM, std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(nrc) : "failed")); }
    since("ncclCommInitRank");
  }
  int rc = build(h, P, st);
  since("build");
  if (rc == 0 && st) rc = upload_state(h, st, true);
  since("upload_state");
  if (rc) { free_handle(h); return rc; }
  *out = h;
  return 0;
}

}  // namespace mpsfm

// ---- C ABI ---------------------------------------------------------------------------------------
// Test hook (tests/test_host_cpu.py; no device involved): takes and gives back `count` host blocks of `bytes` each through
// the block cache twice; returns how many blocks of the second round were recycled ones of the first (by address).
extern "C" int64_t mpsfm_debug_host_cache(int64_t bytes, int32_t count) {
  std::vector<mpsfm::HostBuf<uint8_t>> first((size_t)count), second((size_t)count);
  std::vector<const void*> seen;
  for (auto& b : first) { b.alloc((size_t)bytes); b[0] = 1; b[(size_t)bytes - 1] = 2; seen.push_back(b.data()); }
  first.clear();
  int64_t reused = 0;
  for (auto& b : second) {
    b.alloc((size_t)bytes);
    b[0] = 3;
    reused += std::find(seen.begin(), seen.end(), (const void*)b.data()) != seen.end() ? 1 : 0;
  }
  return reused;
}

// Test hook (tests/test_host_cpu.py; no device involved): `reps` rounds of an `nparts`-part job through the table build's
// worker pool; returns the number of parts that did not run exactly once.
extern "C" int64_t mpsfm_debug_run_parts(int32_t nparts, int32_t reps) {
  int64_t bad = 0;
  for (int r = 0; r < reps; ++r) {
    std::vector<std::atomic<int>> hits((size_t)std::max(nparts, 1));
    for (auto& x : hits) x.store(0);
    mpsfm::run_parts(nparts, [&](int t, int np) {
      if (np != std::max(nparts, 1) || t < 0 || t >= np) return;
      volatile double acc = 0.0;
      for (int k = 0; k < 2000; ++k) acc = acc + (double)k * 1e-9;  // a little work, so that parts overlap
      hits[(size_t)t].fetch_add(1);
    });
    for (auto& x : hits) bad += x.load() == 1 ? 0 : 1;
  }
  return bad;
}

// Test hook (tests/test_dist_cpu.py; no device involved): the camera-graph union of a landmark-sharded run as the ranks
// compute it — `world` adjacency matrices (n x n bytes each) packed per rank, summed like the all-reduce does, unpacked
// into `out` (n x n bytes).  Returns the digits per double used.
extern "C" int mpsfm_debug_graph_union(const uint8_t* adj, int32_t world, int32_t n, uint8_t* out) {
  std::vector<double> sum;
  for (int r = 0; r < world; ++r) {
    mpsfm::CamGraph g;
    g.init(n);
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b)
        if (adj[((size_t)r * n + a) * n + b]) g.set(a, b);
    std::vector<double> packed;
    mpsfm::pack_graph(g, world, packed);
    if (sum.empty()) sum.assign(packed.size(), 0.0);
    for (size_t i = 0; i < packed.size(); ++i) sum[i] += packed[i];
  }
  mpsfm::CamGraph u;
  u.init(n);
  mpsfm::unpack_graph(sum, world, u);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) out[(size_t)a * n + b] = u.get(a, b) ? 1 : 0;
  return mpsfm::graph_digits(world);
}

extern "C" {

int mpsfm_abi_version(void) { return MPSFM_ABI_VERSION; }
const char* mpsfm_last_error(void) { return g_err.c_str(); }

int mpsfm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int good = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
  }
  return good;
}

int mpsfm_comm_unique_id(uint8_t id[128]) {
  if (!id) return fail(MPSFM_EINVAL, "id is NULL");
  Rccl& R = rccl();
  if (!R.ok) return fail(MPSFM_ECOMM, R.why);
  Rccl::UniqueId u;
  const int rc = R.GetUniqueId(&u);
  if (rc != 0) return fail(MPSFM_ECOMM, std::string("ncclGetUniqueId: ") + (R.GetErrorString ? R.GetErrorString(rc) : "failed"));
  std::memcpy(id, u.internal, 128);
  return 0;
}

void mpsfm_ba_default_options(mpsfm_ba_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
}

int mpsfm_ba_create(const mpsfm_ba_problem* problem, const mpsfm_ba_state* initial, const mpsfm_ba_options* options,
                    mpsfm_ba_handle** out) {
  return create_impl(problem, initial, options, out);
}

int mpsfm_ba_set_state(mpsfm_ba_handle* h, const mpsfm_ba_state* state) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  if (hipSetDevice(h->device) != hipSuccess) return fail(MPSFM_EHIP, "hipSetDevice failed");
  return upload_state(h, state, false);
}

int mpsfm_ba_reset_state(mpsfm_ba_handle* h) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(h->d_q, h->d_q0, sizeof(double) * 4 * h->nc, hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_t, h->d_t0, sizeof(double) * 3 * h->nc, hipMemcpyDeviceToDevice, h->stream));
  if (h->np > 0) HIP_TRY(hipMemcpyAsync(h->d_pts, h->d_pts0, sizeof(double) * 3 * h->np, hipMemcpyDeviceToDevice, h->stream));
  h->scales_ready = false;
  return 0;
}

int mpsfm_ba_solve_resident(mpsfm_ba_handle* h, mpsfm_ba_summary* summary) {
  if (!h || !summary) return fail(MPSFM_EINVAL, "handle or summary is NULL");
  return solve_impl(h, summary);
}

int mpsfm_ba_get_state(mpsfm_ba_handle* h, mpsfm_ba_state* st) {
  if (!h || !st) return fail(MPSFM_EINVAL, "handle or state is NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));  // copies on the handle's own stream, never the legacy null stream (see solve_impl)
  if (h->nc > 0) {
    HIP_TRY(hipMemcpyAsync(st->cam_quat_xyzw, h->d_q, sizeof(double) * 4 * h->nc, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(st->cam_t, h->d_t, sizeof(double) * 3 * h->nc, hipMemcpyDeviceToHost, h->stream));
  }
  if (h->d_perm) {  // back into the caller's order on the device, one copy straight into the caller's array
    launch_permute_pts(h->np, h->d_perm, h->d_pts, h->d_user_pts, true, h->stream);
    HIP_TRY(hipMemcpyAsync(st->pts, h->d_user_pts, sizeof(double) * 3 * h->np, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
  }
  std::vector<double> sorted((size_t)h->np * 3);
  if (h->np > 0) HIP_TRY(hipMemcpyAsync(sorted.data(), h->d_pts, sizeof(double) * 3 * h->np, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  parallel_ranges(h->np, 16384, [&](int64_t k0, int64_t k1) {
    for (int64_t k = k0; k < k1; ++k) {
      double* d = st->pts + 3 * (size_t)h->perm[(size_t)k];
      d[0] = sorted[3 * (size_t)k]; d[1] = sorted[3 * (size_t)k + 1]; d[2] = sorted[3 * (size_t)k + 2];
    }
  });
  return 0;
}

void mpsfm_ba_destroy(mpsfm_ba_handle* h) { free_handle(h); }

int mpsfm_ba_solve(const mpsfm_ba_problem* problem, mpsfm_ba_state* state, const mpsfm_ba_options* options,
                   mpsfm_ba_summary* summary) {
  if (!state || !summary) return fail(MPSFM_EINVAL, "state or summary is NULL");
  mpsfm_ba_handle* h = nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!options || options->verbose < 2) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mpsfm_ba] one-shot: %-25s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  int rc = create_impl(problem, state, options, &h);
  if (rc) return rc;
  lap("create");
  rc = solve_impl(h, summary);
  lap("solve");
  if (rc == 0) rc = mpsfm_ba_get_state(h, state);
  lap("get_state");
  free_handle(h);
  lap("destroy");
  return rc;
}

int mpsfm_ba_eval_cost(mpsfm_ba_handle* h, double* cost_reproj, double* cost_depth) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  launch_cam_scales(h->nc, h->d_cam_slot, h->d_cmask, h->d_diagU, 0, h->d_cs, h->stream);
  launch_build_camtab(h->nc, h->d_q, h->d_t, h->d_intr, h->d_intr_idx, h->d_cs, h->d_camtab, h->stream);
  h->scales_ready = false;
  double a[3], b[3];
  if (int rc = cost_of_records(h, h->nrec, h->d_rec_cam, h->d_rec_pt, h->d_rec_meta, h->d_rec_xy, h->d_rec_d, h->d_rec_m, h->d_rec_a, a)) return rc;
  if (int rc = cost_of_records(h, h->nfixed, h->d_fx_cam, h->d_fx_pt, h->d_fx_meta, h->d_fx_xy, h->d_fx_d, h->d_fx_m, h->d_fx_a, b)) return rc;
  if (cost_reproj) *cost_reproj = a[0] + b[0];
  if (cost_depth) *cost_depth = a[1] + b[1];
  return 0;
}

int mpsfm_ba_dense_plan(mpsfm_ba_handle* h, int64_t info[10]) {
  if (!h || !info) return fail(MPSFM_EINVAL, "handle or info is NULL");
  const CholPlan& P = h->plan;
  const bool level = dense_level(&h->ov, &h->lp);
  const bool pinv = level && dense_pinv(h->d_dwork, h->nt, &h->ov, &h->lp) != nullptr;
  const int64_t v[10] = {h->ncv, h->nt, level ? P.nlevels : h->nt, P.nd_depth, pinv ? 1 : 0, (int64_t)P.items.size(), P.products, P.roles, h->sblk_blocks,
                         pinv ? 1 : (level ? P.nlevels : (h->nt + 3) / 4 + 1)};
  for (int i = 0; i < 10; ++i) info[i] = v[i];
  return 0;
}
int mpsfm_ba_reduced_dim(mpsfm_ba_handle* h) { return h ? h->n_user : MPSFM_EINVAL; }

int mpsfm_ba_sweep_once(mpsfm_ba_handle* h, double radius, float* elapsed_ms) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  if (!h->scales_ready) if (int rc = prepare_scales(h)) return rc;
  HIP_TRY(hipMemsetAsync(h->d_red, 0, sizeof(double) * (size_t)h->red_count, h->stream));
  SweepArgs a = sweep_args(h, radius);
  // the three parts of the sweep between events: dense chunks | reduction of their slabs | general chunks and long tracks
  HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  launch_track_sweep_dense(a, h->n_dense, h->stream);
  HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  launch_reduce_slabs(h->d_red_dests, h->n_red_dests, h->d_red_srcs, h->d_slab, h->d_Sblk, h->d_gc, h->d_wv, h->d_diagU, nullptr, h->stream);
  HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  a.chunk0 = h->n_dense;
  launch_track_sweep(a, h->nchunks - h->n_dense, false, h->stream);
  HIP_TRY(hipEventRecord(h->ev[3], h->stream));
  if (h->nchunks + h->nlong > 0) launch_reduce_cols(h->d_part, h->nchunks + h->nlong, 4, 3, 1u << 2, h->d_redsc, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  h->last_radius = radius;
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[3]));
  if (elapsed_ms) *elapsed_ms = ms;
  return 0;
}

// Diagnostics (scripts/dbg_sweep_trace.py): the first `count` 8-byte words of the landmark-diagonal buffer, where the dense sweep leaves
// its phase stamps under debug flag 128.
int mpsfm_debug_read_trace(mpsfm_ba_handle* h, long long* out, int64_t count) {
  if (!h || !out || count < 0 || count > 3 * std::max<int64_t>(h->np, 1)) return fail(MPSFM_EINVAL, "bad trace request");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(out, h->d_diagV, sizeof(long long) * (size_t)count, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// Diagnostics / tests (tests/test_gpu_devbuild.py): table `which` of the handle copied to `out` (at most `cap` bytes); returns the
// table's size in bytes, or a negative error code.  which: 0 chunk headers, 1 chunk cameras, 2 rec_cam, 3 rec_pt, 4 rec_meta, 5 rec_xy,
// 6 rec_d, 7 rec_m, 8 rec_a, 9 pt_rec_start, 10 pt_kv, 11 fx_cam, 12 fx_pt, 13 fx_meta, 14 fx_xy, 15 fx_d, 16 fx_m, 17 fx_a,
// 18 landmark order (host), 19 reduction destinations, 20 reduction sources, 21 camera slots (host), 22: 1 byte, built on the device?,
// 23 blk_desc, 24 blk_ent_start, 25 ents (pair tables of the general chunks)
int64_t mpsfm_debug_table(mpsfm_ba_handle* h, int32_t which, void* out, int64_t cap) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  const void* src = nullptr;
  int64_t bytes = 0;
  bool host = false;
  const int64_t nr = h->nrec, np1 = h->np + 1, nf = h->nfixed;
  switch (which) {
    case 0: src = h->d_chunks; bytes = (int64_t)sizeof(ChunkHdr) * h->nchunks; break;
    case 1: src = h->d_chunk_cams; bytes = 4 * h->n_chunk_cams; break;
    case 2: src = h->d_rec_cam; bytes = 4 * nr; break;
    case 3: src = h->d_rec_pt; bytes = 4 * nr; break;
    case 4: src = h->d_rec_meta; bytes = 4 * nr; break;
    case 5: src = h->d_rec_xy; bytes = 16 * nr; break;
    case 6: src = h->d_rec_d; bytes = 8 * nr; break;
    case 7: src = h->d_rec_m; bytes = 8 * nr; break;
    case 8: src = h->d_rec_a; bytes = 8 * nr; break;
    case 9: src = h->d_pt_rec_start; bytes = 4 * np1; break;
    case 10: src = h->d_pt_kv; bytes = 2 * np1; break;
    case 11: src = h->d_fx_cam; bytes = 4 * nf; break;
    case 12: src = h->d_fx_pt; bytes = 4 * nf; break;
    case 13: src = h->d_fx_meta; bytes = 4 * nf; break;
    case 14: src = h->d_fx_xy; bytes = 16 * nf; break;
    case 15: src = h->d_fx_d; bytes = 8 * nf; break;
    case 16: src = h->d_fx_m; bytes = 8 * nf; break;
    case 17: src = h->d_fx_a; bytes = 8 * nf; break;
    case 18: src = h->perm.data(); bytes = 4 * (int64_t)h->perm.size(); host = true; break;
    case 19: src = h->d_red_dests; bytes = (int64_t)sizeof(RedDest) * h->n_red_dests; break;
    case 20: src = h->d_red_srcs; bytes = 4 * h->n_red_srcs; break;
    case 21: src = h->cam_slot_h.data(); bytes = 4 * (int64_t)h->cam_slot_h.size(); host = true; break;
    case 23: src = h->d_blk_desc; bytes = 4 * h->n_blk_desc; break;
    case 24: src = h->d_blk_ent_start; bytes = 4 * h->n_blk_ent_start; break;
    case 25: src = h->d_ents; bytes = 4 * h->n_ents; break;
    case 26: src = h->d_part; bytes = 32 * (int64_t)h->nchunks; break;
    case 22: { static uint8_t flag; flag = h->built_on_device ? 1 : 0; src = &flag; bytes = 1; host = true; break; }
    default: return fail(MPSFM_EINVAL, "unknown table");
  }
  if (!out || cap < bytes) return bytes;
  if (bytes == 0) return 0;
  if (host) { std::memcpy(out, src, (size_t)bytes); return bytes; }
  if (hipSetDevice(h->device) != hipSuccess) return fail(MPSFM_EHIP, "hipSetDevice failed");
  if (hipMemcpyAsync(out, src, (size_t)bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return fail(MPSFM_EHIP, "copy failed");
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(MPSFM_EHIP, "sync failed");
  return bytes;
}

int mpsfm_ba_sweep_parts(mpsfm_ba_handle* h, float ms[3], int64_t info[4]) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  if (ms)
    for (int k = 0; k < 3; ++k) HIP_TRY(hipEventElapsedTime(&ms[k], h->ev[k], h->ev[k + 1]));
  if (info) { info[0] = h->n_dense; info[1] = h->nchunks - h->n_dense; info[2] = h->nlong; info[3] = h->n_red_dests; }
  return 0;
}

// phase clocks of the last single-launch solve (local_lm.hip), 100 MHz ticks: sweep, barrier 1, dense + cameras, update, barrier 2,
// decision, iterations, then (debug flag 64 << 8) inside the dense phase: assemble, stacked factorisations, their barrier, trailing
// updates, back substitution; returns 0 when the handle does not take that path
int mpsfm_debug_local_clocks(mpsfm_ba_handle* h, int64_t out[12]) {
  if (!h || !h->local_ok) return 0;
  if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return 0;
  if (hipMemcpy(out, h->d_local_sync + 1, sizeof(int64_t) * 12, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return 1;
}
int mpsfm_ba_dense_solve_once(mpsfm_ba_handle* h, float* elapsed_ms) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  if (int rc = run_dense(h, h->last_radius)) return rc;
  HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  HIP_TRY(hipMemsetAsync(h->d_fail, 0, sizeof(int), h->stream));  // no k_cam_update follows here to re-arm the flag
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipGetLastError());
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  if (elapsed_ms) *elapsed_ms = ms;
  return 0;
}

// S and rhs of the last sweep (with the LM damping of its radius), plus the last dense solution
// S and the right-hand side in the CALLER's camera order (6 rows per variable camera), whatever slot order the handle uses.
int mpsfm_ba_get_reduced_system(mpsfm_ba_handle* h, double* S, double* rhs, int32_t n) {
  if (!h) return fail(MPSFM_EINVAL, "handle is NULL");
  if (n != h->n_user) return fail(MPSFM_EINVAL, "n does not match the reduced dimension");
  HIP_TRY(hipSetDevice(h->device));
  std::vector<double> red((size_t)h->red_count);
  HIP_TRY(hipMemcpyAsync(red.data(), h->d_red, sizeof(double) * red.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double* Sb = red.data(); const double* gc = Sb + h->sblk_count; const double* wv = gc + h->n_user; const double* dU = wv + h->n_user;
  const mpsfm_ba_options& o = h->opt;
  const BlockSky sky{h->sky_first.data(), h->sky_start.data(), h->sky_index.empty() ? nullptr : h->sky_index.data(), h->ncv};
  for (int R = 0; R < n; ++R)
    for (int C = 0; C < n; ++C) {
      const int br = h->nat_slot[(size_t)(R / 6)], a = R % 6, bc = h->nat_slot[(size_t)(C / 6)], b = C % 6;
      double v;
      const int lo = std::min(br, bc), hi = std::max(br, bc);
      if (!sky_has(sky, lo, hi)) v = 0.0;
      else if (br < bc) v = Sb[sky_block(sky, br, bc) * 36 + a * 6 + b];
      else if (br > bc) v = Sb[sky_block(sky, bc, br) * 36 + b * 6 + a];
      else v = Sb[sky_block(sky, br, br) * 36 + (a <= b ? a * 6 + b : b * 6 + a)];
      if (R == C) v += std::min(std::max(dU[6 * br + a], o.min_lm_diagonal), o.max_lm_diagonal) / h->last_radius;
      if (S) S[(size_t)R * n + C] = v;
    }
  if (rhs) for (int i = 0; i < n; ++i) { const int q = 6 * h->nat_slot[(size_t)(i / 6)] + i % 6; rhs[i] = wv[q] - gc[q]; }
  return 0;
}

int mpsfm_ba_get_dense_solution(mpsfm_ba_handle* h, double* y, int32_t n) {
  if (!h || !y) return fail(MPSFM_EINVAL, "handle or y is NULL");
  if (n != h->n_user) return fail(MPSFM_EINVAL, "n does not match the reduced dimension");
  HIP_TRY(hipSetDevice(h->device));
  std::vector<double> ys((size_t)std::max(h->n_user, 1));
  HIP_TRY(hipMemcpyAsync(ys.data(), h->d_yc, sizeof(double) * (size_t)h->n_user, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) y[i] = ys[(size_t)(6 * h->nat_slot[(size_t)(i / 6)] + i % 6)];
  return 0;
}

}  // extern "C"
