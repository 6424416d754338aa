// Row f2: the track-graph logic of COLMAP's IncrementalTriangulator that MpsfmTriangulator drives
// (reference mpsfm/sfm/mapper/triangulator.py:32-48, 88-100, 123, 165-175; mapper/base.py:434, 448, 482-485) —
// Find / Create / Continue, Complete, Merge and Retriangulate(ignore_image_ids) — as host C++ behind the C ABI, with the
// candidate tracks of a call estimated in ONE GPU batch.
//
// The fork's C++ is not in the reference tree: parity unpinned.  What is restated is upstream COLMAP 3.11
// (src/colmap/sfm/incremental_triangulator.cc, scene/observation_manager.cc, estimators/triangulation.cc): the control
// flow below follows it statement by statement in its effects (which observations join which point, in which order);
// the per-track arithmetic is tri_math.h.
//
// GPU batching with sequential semantics.  COLMAP walks the keypoints of an image one by one and every Create / Continue
// changes which observations already have a 3-D point — the input of later keypoints.  The engine therefore first
// collects the candidate track of EVERY keypoint against the state at the start of the call, estimates all of them in
// one launch of k_tri_ransac (one thread per candidate: LORANSAC over lexicographic pairs), and then commits keypoint by
// keypoint in COLMAP's order: a candidate whose observation set is still what it was when the batch ran takes the
// batch result, one that an earlier commit has touched is re-estimated on the spot (same arithmetic, host side).
// The scene itself stays with the caller: every call returns an operation log (add point / add observation / delete
// point) that the Python layer replays on the reference's ObservationManager.
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"
#include "tri_math.h"

namespace mpsfm {
extern thread_local std::string g_err;
int staged_upload(void* dst, const void* src, size_t bytes);
int staged_drain();
static int gfail(int code, const std::string& m) { g_err = m; return code; }

struct TriCand {  // one candidate track handed to the batch kernel
  int64_t v0;     // first view in the view array
  int32_t n;
  TriRansacOptions opt;
};
struct TriResult { double X[3]; uint64_t mask; int32_t ok; int32_t pad; };

__global__ __launch_bounds__(64) void k_tri_ransac(const TriCand* cands, const TriView* views, int n_cands, TriResult* out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_cands) return;
  const TriCand c = cands[i];
  TriResult r{};
  uint64_t m[(kTriMaxViews + 63) / 64];
  r.ok = tri_ransac(views + c.v0, c.n, c.opt, r.X, m) ? 1 : 0;
  r.mask = m[0];
  out[i] = r;
}
}  // namespace mpsfm

using namespace mpsfm;

enum { OP_ADD_POINT = 0, OP_ADD_OBS = 1, OP_DELETE_POINT = 2 };

struct mpsfm_triangulator {
  int device = 0;
  // static: keypoints and correspondence graph
  int n_images = 0;
  std::vector<int64_t> kp_start;
  std::vector<double> kp_xy, intr;
  std::vector<int32_t> kp_image;
  std::vector<int64_t> corr_start, corr_kp;
  // state
  std::vector<uint8_t> registered;
  std::vector<double> R, t;
  std::vector<int64_t> kp_pt;  // engine point index or -1
  struct Pt { double X[3]; std::vector<int64_t> els; bool alive; };
  std::vector<Pt> pts;
  std::unordered_map<int64_t, std::unordered_set<int64_t>> merge_trials;
  std::unordered_map<uint64_t, int> re_num_trials;
  // operation log of the last call
  struct Op { int32_t type; int64_t a, b; double X[3]; };  // ADD_POINT: b = first element in op_els, track length = next op's ... see get_ops
  std::vector<Op> ops;
  std::vector<int64_t> op_els;  // elements of the points added by the logged operations, in order
  bool use_gpu = true;
  int64_t n_batch = 0, n_batch_hits = 0, n_host_estimates = 0;

  int64_t n_kp() const { return (int64_t)kp_image.size(); }
  TriView view(int64_t kp) const {
    TriView v;
    const int im = kp_image[(size_t)kp];
    tri_make_view(&R[9 * (size_t)im], &t[3 * (size_t)im], &intr[4 * (size_t)im], &kp_xy[2 * (size_t)kp], v);
    return v;
  }
  bool has_pt(int64_t kp) const { return kp_pt[(size_t)kp] >= 0; }

  // ---- bookkeeping (ObservationManager::AddPoint3D / AddObservation / DeletePoint3D / MergePoints3D) -----------------
  int64_t add_point(const double* X, const std::vector<int64_t>& els) {
    Pt p; p.X[0] = X[0]; p.X[1] = X[1]; p.X[2] = X[2]; p.els = els; p.alive = true;
    pts.push_back(p);
    const int64_t id = (int64_t)pts.size() - 1;
    for (int64_t kp : els) kp_pt[(size_t)kp] = id;
    ops.push_back(Op{OP_ADD_POINT, id, (int64_t)els.size(), {X[0], X[1], X[2]}});
    op_els.insert(op_els.end(), els.begin(), els.end());
    return id;
  }
  void add_obs(int64_t id, int64_t kp) {
    pts[(size_t)id].els.push_back(kp);
    kp_pt[(size_t)kp] = id;
    ops.push_back(Op{OP_ADD_OBS, id, kp, {0, 0, 0}});
  }
  void delete_point(int64_t id) {
    for (int64_t kp : pts[(size_t)id].els) kp_pt[(size_t)kp] = -1;
    pts[(size_t)id].alive = false;
    pts[(size_t)id].els.clear();
    ops.push_back(Op{OP_DELETE_POINT, id, 0, {0, 0, 0}});
  }
  int64_t merge_points(int64_t a, int64_t b) {
    const double la = (double)pts[(size_t)a].els.size(), lb = (double)pts[(size_t)b].els.size();
    double X[3];
    for (int k = 0; k < 3; ++k) X[k] = (la * pts[(size_t)a].X[k] + lb * pts[(size_t)b].X[k]) / (la + lb);
    std::vector<int64_t> els = pts[(size_t)a].els;
    els.insert(els.end(), pts[(size_t)b].els.begin(), pts[(size_t)b].els.end());
    delete_point(a);
    delete_point(b);
    return add_point(X, els);
  }

  // ---- Find: correspondences of a keypoint in registered images (max_transitivity 1: the direct ones) -----------------
  size_t find(int64_t kp, std::vector<int64_t>& corrs) const {
    corrs.clear();
    size_t num_tri = 0;
    for (int64_t e = corr_start[(size_t)kp]; e < corr_start[(size_t)kp + 1]; ++e) {
      const int64_t c = corr_kp[(size_t)e];
      if (!registered[(size_t)kp_image[(size_t)c]]) continue;
      corrs.push_back(c);
      if (has_pt(c)) ++num_tri;
    }
    return num_tri;
  }
  bool is_two_view_observation(int64_t kp) const {  // CorrespondenceGraph::IsTwoViewObservation
    if (corr_start[(size_t)kp + 1] - corr_start[(size_t)kp] != 1) return false;
    const int64_t c = corr_kp[(size_t)corr_start[(size_t)kp]];
    return corr_start[(size_t)c + 1] - corr_start[(size_t)c] == 1;
  }
  // EstimateTriangulationOptions as Create sets them up: angular residual, exhaustive sampling up to 15 observations
  static TriRansacOptions ransac_options(const mpsfm_tri_options& o, size_t n) {
    TriRansacOptions r{};
    r.min_tri_angle = o.min_angle * M_PI / 180.0;
    r.max_error = o.create_max_angle_error * M_PI / 180.0;
    r.confidence = 0.9999;
    r.max_num_trials = 10000;
    r.min_num_trials = n <= 15 ? (int64_t)n * ((int64_t)n - 1) / 2 : 0;
    r.residual_type = TRI_RESIDUAL_ANGULAR;
    return r;
  }
  // ... and as CompleteImage does: REPROJECTION_ERROR with complete_max_reproj_error in pixels; ONE options object serves
  // the whole keypoint loop there, so the exhaustive-sampling floor set for a short track stays in force for the
  // following longer ones (`sticky`: the value the loop carries, updated here)
  static TriRansacOptions complete_options(const mpsfm_tri_options& o, size_t n, int64_t& sticky) {
    TriRansacOptions r{};
    r.min_tri_angle = o.min_angle * M_PI / 180.0;
    r.max_error = o.complete_max_reproj_error;
    r.confidence = 0.9999;
    r.max_num_trials = 10000;
    if (n <= 15) sticky = (int64_t)n * ((int64_t)n - 1) / 2;
    r.min_num_trials = sticky;
    r.residual_type = TRI_RESIDUAL_REPROJECTION;
    return r;
  }
  static bool same_options(const TriRansacOptions& a, const TriRansacOptions& b) {
    return a.min_tri_angle == b.min_tri_angle && a.max_error == b.max_error && a.confidence == b.confidence &&
           a.max_num_trials == b.max_num_trials && a.min_num_trials == b.min_num_trials && a.residual_type == b.residual_type;
  }
  // the estimator on the host for any track length (heap scratch); inlier flags per view
  static bool host_ransac(const std::vector<TriView>& views, const TriRansacOptions& opt, double* X, std::vector<uint8_t>& inl) {
    const int n = (int)views.size();
    std::vector<double> res((size_t)n), res2((size_t)n);
    std::vector<int> idx((size_t)n);
    std::vector<uint64_t> m((size_t)(n + 63) / 64 + 1);
    inl.assign((size_t)n, 0);
    if (!tri_ransac_scratch(views.data(), n, opt, X, m.data(), res.data(), res2.data(), idx.data())) return false;
    for (int i = 0; i < n; ++i) inl[(size_t)i] = (uint8_t)(m[(size_t)i / 64] >> (i % 64) & 1);
    return true;
  }

  // ---- the batch: candidate sets estimated on the GPU, looked up at commit time ----------------------------------------
  struct Batch {
    std::unordered_map<int64_t, int64_t> of_ref;  // reference keypoint -> candidate index
    std::vector<std::vector<int64_t>> sets;
    std::vector<TriRansacOptions> opts;  // the estimator options each set was estimated with
    std::vector<TriResult> results;
    void add(int64_t ref, const std::vector<int64_t>& set, const TriRansacOptions& opt) {
      of_ref[ref] = (int64_t)sets.size();
      sets.push_back(set);
      opts.push_back(opt);
    }
  };
  int run_batch(Batch& B) {
    const size_t nc = B.sets.size();
    B.results.assign(nc, TriResult{});
    if (nc == 0) return 0;
    std::vector<TriCand> cands(nc);
    std::vector<TriView> views;
    for (size_t i = 0; i < nc; ++i) {
      cands[i].v0 = (int64_t)views.size(); cands[i].n = (int32_t)B.sets[i].size(); cands[i].opt = B.opts[i];
      for (int64_t kp : B.sets[i]) views.push_back(view(kp));
    }
    n_batch += (int64_t)nc;
    if (!use_gpu) {  // MPSFM_TRI_HOST_BATCH=1, a TEST switch: the batch arithmetic on the host, to compare the kernel with
      for (size_t i = 0; i < nc; ++i) {
        uint64_t m[(kTriMaxViews + 63) / 64];
        B.results[i].ok = tri_ransac(views.data() + cands[i].v0, cands[i].n, cands[i].opt, B.results[i].X, m) ? 1 : 0;
        B.results[i].mask = m[0];
      }
      return 0;
    }
    if (hipSetDevice(device) != hipSuccess) return gfail(MPSFM_EHIP, "hipSetDevice failed");
    TriCand* d_c = (TriCand*)cached_malloc(sizeof(TriCand) * nc);
    TriView* d_v = (TriView*)cached_malloc(sizeof(TriView) * views.size());
    TriResult* d_r = (TriResult*)cached_malloc(sizeof(TriResult) * nc);
    int rc = 0;
    if (!d_c || !d_v || !d_r) rc = gfail(MPSFM_ENOMEM, "hipMalloc failed");
    if (!rc) rc = staged_upload(d_c, cands.data(), sizeof(TriCand) * nc);
    if (!rc) rc = staged_upload(d_v, views.data(), sizeof(TriView) * views.size());
    if (!rc) rc = staged_drain();
    hipStream_t st = nullptr;  // a pooled non-blocking stream, never the legacy null stream (see DevBuf in tri_kernels.hip)
    if (!rc && pooled_stream(&st) != hipSuccess) rc = gfail(MPSFM_EHIP, "hipStreamCreate failed");
    if (!rc) {
      hipLaunchKernelGGL(k_tri_ransac, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, st, d_c, d_v, (int)nc, d_r);
      if (hipMemcpyAsync(B.results.data(), d_r, sizeof(TriResult) * nc, hipMemcpyDeviceToHost, st) != hipSuccess) rc = gfail(MPSFM_EHIP, "reading the RANSAC batch back failed");
    }
    if (st) { (void)hipStreamSynchronize(st); release_stream(st); }
    cached_free(d_c); cached_free(d_v); cached_free(d_r);
    return rc;
  }
  // estimate a candidate set: the batch result when set and options are what the batch ran with, else on the spot (host;
  // also every set longer than kTriMaxViews, which the batch never takes)
  bool estimate(const TriRansacOptions& opt, const Batch* B, int64_t ref, const std::vector<int64_t>& set, double* X, std::vector<uint8_t>& inl) {
    if (B) {
      auto it = B->of_ref.find(ref);
      if (it != B->of_ref.end() && B->sets[(size_t)it->second] == set && same_options(B->opts[(size_t)it->second], opt)) {
        const TriResult& r = B->results[(size_t)it->second];
        ++n_batch_hits;
        if (!r.ok) return false;
        X[0] = r.X[0]; X[1] = r.X[1]; X[2] = r.X[2];
        inl.assign(set.size(), 0);
        for (size_t i = 0; i < set.size(); ++i) inl[i] = (uint8_t)(r.mask >> i & 1);
        return true;
      }
    }
    ++n_host_estimates;
    std::vector<TriView> views;
    for (int64_t kp : set) views.push_back(view(kp));
    return host_ransac(views, opt, X, inl);
  }

  // ---- Create / Continue (incremental_triangulator.cc) ------------------------------------------------------------------
  size_t create(const mpsfm_tri_options& o, const Batch* B, int64_t ref, const std::vector<int64_t>& corrs_data) {
    std::vector<int64_t> set;
    for (int64_t kp : corrs_data) if (!has_pt(kp)) set.push_back(kp);
    if (set.size() < 2) return 0;
    if (o.ignore_two_view_tracks && set.size() == 2 && is_two_view_observation(set[0])) return 0;
    double X[3]; std::vector<uint8_t> inl;
    if (!estimate(ransac_options(o, set.size()), B, ref, set, X, inl)) return 0;
    std::vector<int64_t> els;
    for (size_t i = 0; i < set.size(); ++i) if (inl[i]) els.push_back(set[i]);
    add_point(X, els);
    const size_t kMinRecursiveTrackLength = 3;
    if (set.size() - els.size() >= kMinRecursiveTrackLength) return els.size() + create(o, nullptr, ref, set);
    return els.size();
  }
  size_t continue_(double max_angle_error_deg, int64_t ref, const std::vector<int64_t>& corrs) {
    if (has_pt(ref)) return 0;
    double best = DBL_MAX; int64_t best_kp = -1;
    const TriView rv = view(ref);
    for (int64_t kp : corrs) {
      if (!has_pt(kp)) continue;
      const double e = tri_angular_error(rv.xn, pts[(size_t)kp_pt[(size_t)kp]].X, rv.P);
      if (e < best) { best = e; best_kp = kp; }
    }
    if (best_kp >= 0 && best <= max_angle_error_deg * M_PI / 180.0) { add_obs(kp_pt[(size_t)best_kp], ref); return 1; }
    return 0;
  }

  int triangulate_image(const mpsfm_tri_options& o, int im, int64_t* count) {
    *count = 0;
    if (!registered[(size_t)im]) return 0;
    std::vector<int64_t> corrs;
    Batch B;
    for (int64_t kp = kp_start[(size_t)im]; kp < kp_start[(size_t)im + 1]; ++kp) {  // candidates against the state now
      find(kp, corrs);
      if (corrs.empty()) continue;
      std::vector<int64_t> set;
      for (int64_t c : corrs) if (!has_pt(c)) set.push_back(c);
      if (!has_pt(kp)) set.push_back(kp);
      if (set.size() < 2 || set.size() > (size_t)kTriMaxViews) continue;  // longer ones: host estimate at commit time
      if (o.ignore_two_view_tracks && set.size() == 2 && is_two_view_observation(set[0])) continue;
      B.add(kp, set, ransac_options(o, set.size()));
    }
    if (int rc = run_batch(B)) return rc;
    for (int64_t kp = kp_start[(size_t)im]; kp < kp_start[(size_t)im + 1]; ++kp) {  // commit in COLMAP's order
      const size_t num_tri = find(kp, corrs);
      if (corrs.empty()) continue;
      if (num_tri == 0) {
        corrs.push_back(kp);
        *count += (int64_t)create(o, &B, kp, corrs);
      } else {
        *count += (int64_t)continue_(o.continue_max_angle_error, kp, corrs);
        corrs.push_back(kp);
        *count += (int64_t)create(o, &B, kp, corrs);
      }
    }
    return 0;
  }

  // ---- CompleteImage: new points from keypoints none of whose correspondences is triangulated, existing tracks completed
  int complete_image(const mpsfm_tri_options& o, int im, int64_t* count) {
    *count = 0;
    if (!registered[(size_t)im]) return 0;
    std::vector<int64_t> corrs;
    Batch B;
    int64_t sticky = 0;  // predicted course of the loop's min_num_trials (see complete_options)
    for (int64_t kp = kp_start[(size_t)im]; kp < kp_start[(size_t)im + 1]; ++kp) {
      if (has_pt(kp) || (o.ignore_two_view_tracks && is_two_view_observation(kp))) continue;
      if (find(kp, corrs) || corrs.empty()) continue;
      corrs.push_back(kp);
      const TriRansacOptions opt = complete_options(o, corrs.size(), sticky);
      if (corrs.size() <= (size_t)kTriMaxViews) B.add(kp, corrs, opt);
    }
    if (int rc = run_batch(B)) return rc;
    sticky = 0;
    for (int64_t kp = kp_start[(size_t)im]; kp < kp_start[(size_t)im + 1]; ++kp) {
      if (has_pt(kp)) { *count += (int64_t)complete(o, kp_pt[(size_t)kp]); continue; }
      if (o.ignore_two_view_tracks && is_two_view_observation(kp)) continue;
      if (find(kp, corrs) || corrs.empty()) continue;
      corrs.push_back(kp);
      double X[3]; std::vector<uint8_t> inl;
      if (!estimate(complete_options(o, corrs.size(), sticky), &B, kp, corrs, X, inl)) continue;
      std::vector<int64_t> els;
      for (size_t i = 0; i < corrs.size(); ++i) if (inl[i]) els.push_back(corrs[i]);
      *count += (int64_t)els.size();
      add_point(X, els);
    }
    return 0;
  }

  // ---- Complete / Merge ---------------------------------------------------------------------------------------------------
  size_t complete(const mpsfm_tri_options& o, int64_t id) {
    size_t num = 0;
    if (id < 0 || id >= (int64_t)pts.size() || !pts[(size_t)id].alive) return 0;
    const double max_sq = o.complete_max_reproj_error * o.complete_max_reproj_error;
    std::vector<int64_t> queue = pts[(size_t)id].els;
    for (int tr = 0; tr < o.complete_max_transitivity; ++tr) {
      if (queue.empty()) break;
      const std::vector<int64_t> prev = queue;
      queue.clear();
      for (int64_t q : prev)
        for (int64_t e = corr_start[(size_t)q]; e < corr_start[(size_t)q + 1]; ++e) {
          const int64_t c = corr_kp[(size_t)e];
          const int im = kp_image[(size_t)c];
          if (!registered[(size_t)im] || has_pt(c)) continue;
          const TriView v = view(c);
          if (tri_sq_reproj_error(&kp_xy[2 * (size_t)c], pts[(size_t)id].X, v.P, &intr[4 * (size_t)im]) > max_sq) continue;
          add_obs(id, c);
          if (tr < o.complete_max_transitivity - 1) queue.push_back(c);
          ++num;
        }
    }
    return num;
  }
  size_t merge(const mpsfm_tri_options& o, int64_t id) {
    if (id < 0 || id >= (int64_t)pts.size() || !pts[(size_t)id].alive) return 0;
    const double max_sq = o.merge_max_reproj_error * o.merge_max_reproj_error;
    const std::vector<int64_t> els = pts[(size_t)id].els;  // the point may be replaced below
    for (int64_t el : els)
      for (int64_t e = corr_start[(size_t)el]; e < corr_start[(size_t)el + 1]; ++e) {
        const int64_t c = corr_kp[(size_t)e];
        if (!registered[(size_t)kp_image[(size_t)c]]) continue;
        const int64_t other = kp_pt[(size_t)c];
        if (other < 0 || other == id || merge_trials[id].count(other)) continue;
        merge_trials[id].insert(other);
        merge_trials[other].insert(id);
        const Pt &A = pts[(size_t)id], &Bp = pts[(size_t)other];
        const double la = (double)A.els.size(), lb = (double)Bp.els.size();
        double X[3];
        for (int k = 0; k < 3; ++k) X[k] = (la * A.X[k] + lb * Bp.X[k]) / (la + lb);
        bool ok = true;
        for (const Pt* P : {&A, &Bp}) {
          for (int64_t kp : P->els) {
            const TriView v = view(kp);
            if (tri_sq_reproj_error(&kp_xy[2 * (size_t)kp], X, v.P, &intr[4 * (size_t)kp_image[(size_t)kp]]) > max_sq) { ok = false; break; }
          }
          if (!ok) break;
        }
        if (ok) {
          const size_t num_merged = A.els.size() + Bp.els.size();
          const int64_t merged = merge_points(id, other);
          const size_t rec = merge(o, merged);
          return rec > 0 ? rec : num_merged;
        }
      }
    return 0;
  }

  // ---- Retriangulate (with the fork's ignore_image_ids) ------------------------------------------------------------------
  int retriangulate(const mpsfm_tri_options& o, const std::unordered_set<int>& ignore, int64_t* count) {
    *count = 0;
    // image pairs with their correspondence and triangulated-correspondence counts (ObservationManager::ImagePairs)
    struct PairStat { int64_t total = 0, tri = 0; std::vector<std::pair<int64_t, int64_t>> corrs; };
    std::unordered_map<uint64_t, PairStat> pairs;
    std::vector<uint64_t> order;
    for (int64_t kp = 0; kp < n_kp(); ++kp)
      for (int64_t e = corr_start[(size_t)kp]; e < corr_start[(size_t)kp + 1]; ++e) {
        const int64_t c = corr_kp[(size_t)e];
        const int i1 = kp_image[(size_t)kp], i2 = kp_image[(size_t)c];
        if (i1 >= i2) continue;  // every correspondence once, smaller image first
        const uint64_t key = ((uint64_t)(uint32_t)i1 << 32) | (uint32_t)i2;
        auto ins = pairs.emplace(key, PairStat{});
        if (ins.second) order.push_back(key);
        PairStat& S = ins.first->second;
        S.total++;
        if (kp_pt[(size_t)kp] >= 0 && kp_pt[(size_t)kp] == kp_pt[(size_t)c]) S.tri++;
        S.corrs.emplace_back(kp, c);
      }
    std::sort(order.begin(), order.end());
    // COLMAP decides pair by pair on the LIVE ratio (num_tri_corrs moves with every Create / Continue of earlier pairs) and
    // spends a trial only on a pair it works on.  The batch needs its candidates up front: it takes the pairs that are
    // under-reconstructed NOW (ratios only grow during the call, so this is a superset of what the walk will take).
    auto eligible = [&](uint64_t key, double ratio) {
      if (ratio >= o.re_min_ratio) return false;
      const int i1 = (int)(key >> 32), i2 = (int)(key & 0xffffffffu);
      if (ignore.count(i1) || ignore.count(i2)) return false;
      if (!registered[(size_t)i1] || !registered[(size_t)i2]) return false;
      auto it = re_num_trials.find(key);
      return !(it != re_num_trials.end() && it->second >= o.re_max_trials);
    };
    Batch B;
    std::vector<uint64_t> todo;
    for (uint64_t key : order) {
      const PairStat& S = pairs[key];
      if (!eligible(key, (double)S.tri / (double)S.total)) continue;
      todo.push_back(key);
      for (const auto& pr : S.corrs)
        if (!has_pt(pr.first) && !has_pt(pr.second)) {
          if (o.ignore_two_view_tracks && is_two_view_observation(pr.first)) continue;
          if (B.of_ref.count(pr.first)) continue;
          B.add(pr.first, {pr.first, pr.second}, ransac_options(o, 2));
        }
    }
    if (int rc = run_batch(B)) return rc;
    for (uint64_t key : todo) {
      const PairStat& S = pairs[key];
      int64_t tri = 0;  // live num_tri_corrs of the pair
      for (const auto& pr : S.corrs)
        if (kp_pt[(size_t)pr.first] >= 0 && kp_pt[(size_t)pr.first] == kp_pt[(size_t)pr.second]) ++tri;
      if (!eligible(key, (double)tri / (double)S.total)) continue;
      re_num_trials[key] += 1;
      for (const auto& pr : S.corrs) {
        const bool h1 = has_pt(pr.first), h2 = has_pt(pr.second);
        if (h1 && h2) continue;
        if (h1 && !h2) *count += (int64_t)continue_(o.re_max_angle_error, pr.second, {pr.first});
        else if (!h1 && h2) *count += (int64_t)continue_(o.re_max_angle_error, pr.first, {pr.second});
        else *count += (int64_t)create(o, &B, pr.first, {pr.first, pr.second});
      }
    }
    return 0;
  }
};

static int check_engine(mpsfm_triangulator* h) { return h ? 0 : gfail(MPSFM_EINVAL, "triangulator handle is NULL"); }

extern "C" {

void mpsfm_tri_default_options(mpsfm_tri_options* o) {
  if (!o) return;
  o->max_transitivity = 1; o->create_max_angle_error = 2.0; o->continue_max_angle_error = 2.0; o->merge_max_reproj_error = 4.0;
  o->complete_max_reproj_error = 4.0; o->complete_max_transitivity = 5; o->re_max_angle_error = 5.0; o->re_min_ratio = 0.2;
  o->re_max_trials = 1; o->min_angle = 1.5; o->ignore_two_view_tracks = 1;
}

int mpsfm_triangulator_create(const mpsfm_tri_graph* g, int32_t device, mpsfm_triangulator** out) {
  if (!g || !out) return gfail(MPSFM_EINVAL, "NULL argument");
  *out = nullptr;
  if (g->n_images < 0 || (g->n_images > 0 && (!g->kp_start || !g->cam_intr))) return gfail(MPSFM_EINVAL, "image arrays are NULL");
  const int64_t nkp = g->n_images > 0 ? g->kp_start[g->n_images] : 0;
  if (nkp < 0 || (nkp > 0 && (!g->kp_xy || !g->corr_start))) return gfail(MPSFM_EINVAL, "keypoint arrays are NULL");
  for (int i = 0; i < g->n_images; ++i) if (g->kp_start[i + 1] < g->kp_start[i]) return gfail(MPSFM_EINVAL, "kp_start must be non-decreasing");
  const int64_t nco = nkp > 0 ? g->corr_start[nkp] : 0;
  if (nco > 0 && !g->corr_kp) return gfail(MPSFM_EINVAL, "corr_kp is NULL");
  for (int64_t k = 0; k < nkp; ++k) if (g->corr_start[k + 1] < g->corr_start[k]) return gfail(MPSFM_EINVAL, "corr_start must be non-decreasing");
  for (int64_t e = 0; e < nco; ++e) if (g->corr_kp[e] < 0 || g->corr_kp[e] >= nkp) return gfail(MPSFM_EINVAL, "correspondence out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gfail(MPSFM_ENODEVICE, "no HIP device visible: libmpsfm_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return gfail(MPSFM_EINVAL, "device ordinal out of range");
  if (device >= kMaxDevices) return gfail(MPSFM_EUNSUPPORTED, "device ordinals beyond 15 are not supported (per-device pools)");
  auto* h = new mpsfm_triangulator();
  h->device = device;
  h->n_images = g->n_images;
  h->kp_start.assign(g->kp_start, g->kp_start + g->n_images + 1);
  h->kp_xy.assign(g->kp_xy, g->kp_xy + 2 * nkp);
  h->intr.assign(g->cam_intr, g->cam_intr + 4 * (size_t)g->n_images);
  h->corr_start.assign(g->corr_start, g->corr_start + nkp + 1);
  h->corr_kp.assign(g->corr_kp, g->corr_kp + nco);
  h->kp_image.resize((size_t)nkp);
  for (int i = 0; i < g->n_images; ++i) for (int64_t k = g->kp_start[i]; k < g->kp_start[i + 1]; ++k) h->kp_image[(size_t)k] = i;
  h->registered.assign((size_t)g->n_images, 0);
  h->R.assign(9 * (size_t)g->n_images, 0.0); h->t.assign(3 * (size_t)g->n_images, 0.0);
  h->kp_pt.assign((size_t)nkp, -1);
  if (const char* e = std::getenv("MPSFM_TRI_HOST_BATCH")) h->use_gpu = std::atoi(e) == 0;  // tests: batch arithmetic on the host
  *out = h;
  return 0;
}

void mpsfm_triangulator_destroy(mpsfm_triangulator* h) { delete h; }

int mpsfm_triangulator_set_state(mpsfm_triangulator* h, const mpsfm_tri_state* s) {
  if (int rc = check_engine(h)) return rc;
  if (!s || (h->n_images > 0 && (!s->registered || !s->cam_quat_xyzw || !s->cam_t)) || (h->n_kp() > 0 && !s->kp_point) || (s->n_points > 0 && !s->xyz))
    return gfail(MPSFM_EINVAL, "state arrays are NULL");
  for (int i = 0; i < h->n_images; ++i) {
    h->registered[(size_t)i] = s->registered[i];
    quat_to_R(s->cam_quat_xyzw + 4 * i, &h->R[9 * (size_t)i]);
    for (int k = 0; k < 3; ++k) h->t[3 * (size_t)i + k] = s->cam_t[3 * i + k];
  }
  h->pts.clear();
  h->pts.resize((size_t)s->n_points);
  for (int64_t p = 0; p < s->n_points; ++p) {
    h->pts[(size_t)p].alive = true;
    for (int k = 0; k < 3; ++k) h->pts[(size_t)p].X[k] = s->xyz[3 * p + k];
  }
  for (int64_t k = 0; k < h->n_kp(); ++k) {
    const int64_t p = s->kp_point[k];
    if (p >= s->n_points) return gfail(MPSFM_EINVAL, "kp_point out of range");
    h->kp_pt[(size_t)k] = p < 0 ? -1 : p;
    if (p >= 0) h->pts[(size_t)p].els.push_back(k);
  }
  h->merge_trials.clear();
  h->ops.clear();
  h->op_els.clear();
  return 0;
}

static void begin_call(mpsfm_triangulator* h) { h->ops.clear(); h->op_els.clear(); }

int mpsfm_triangulator_triangulate_image(mpsfm_triangulator* h, const mpsfm_tri_options* o, int32_t image, int64_t* count) {
  if (int rc = check_engine(h)) return rc;
  if (!o || !count || image < 0 || image >= h->n_images) return gfail(MPSFM_EINVAL, "bad argument");
  begin_call(h);
  return h->triangulate_image(*o, image, count);
}

int mpsfm_triangulator_complete_image(mpsfm_triangulator* h, const mpsfm_tri_options* o, int32_t image, int64_t* count) {
  if (int rc = check_engine(h)) return rc;
  if (!o || !count || image < 0 || image >= h->n_images) return gfail(MPSFM_EINVAL, "bad argument");
  begin_call(h);
  return h->complete_image(*o, image, count);
}

int mpsfm_triangulator_complete_tracks(mpsfm_triangulator* h, const mpsfm_tri_options* o, const int64_t* points, int64_t n, int64_t* count) {
  if (int rc = check_engine(h)) return rc;
  if (!o || !count || (n > 0 && !points)) return gfail(MPSFM_EINVAL, "bad argument");
  begin_call(h);
  *count = 0;
  if (n < 0) { for (int64_t p = 0, np = (int64_t)h->pts.size(); p < np; ++p) *count += (int64_t)h->complete(*o, p); }  // all tracks
  else for (int64_t i = 0; i < n; ++i) *count += (int64_t)h->complete(*o, points[i]);
  return 0;
}

int mpsfm_triangulator_merge_tracks(mpsfm_triangulator* h, const mpsfm_tri_options* o, const int64_t* points, int64_t n, int64_t* count) {
  if (int rc = check_engine(h)) return rc;
  if (!o || !count || (n > 0 && !points)) return gfail(MPSFM_EINVAL, "bad argument");
  begin_call(h);
  *count = 0;
  if (n < 0) { for (int64_t p = 0, np = (int64_t)h->pts.size(); p < np; ++p) *count += (int64_t)h->merge(*o, p); }
  else for (int64_t i = 0; i < n; ++i) *count += (int64_t)h->merge(*o, points[i]);
  return 0;
}

int mpsfm_triangulator_retriangulate(mpsfm_triangulator* h, const mpsfm_tri_options* o, const int32_t* ignore_images, int32_t n_ignore, int64_t* count) {
  if (int rc = check_engine(h)) return rc;
  if (!o || !count || (n_ignore > 0 && !ignore_images)) return gfail(MPSFM_EINVAL, "bad argument");
  begin_call(h);
  std::unordered_set<int> ig(ignore_images, ignore_images + (n_ignore > 0 ? n_ignore : 0));
  return h->retriangulate(*o, ig, count);
}

int64_t mpsfm_triangulator_num_ops(mpsfm_triangulator* h) { return h ? (int64_t)h->ops.size() : MPSFM_EINVAL; }
int64_t mpsfm_triangulator_num_points(mpsfm_triangulator* h) { return h ? (int64_t)h->pts.size() : MPSFM_EINVAL; }

int mpsfm_triangulator_get_ops(mpsfm_triangulator* h, int32_t* type, int64_t* a, int64_t* b, double* xyz) {
  if (int rc = check_engine(h)) return rc;
  for (size_t i = 0; i < h->ops.size(); ++i) {
    type[i] = h->ops[i].type; a[i] = h->ops[i].a; b[i] = h->ops[i].b;
    xyz[3 * i] = h->ops[i].X[0]; xyz[3 * i + 1] = h->ops[i].X[1]; xyz[3 * i + 2] = h->ops[i].X[2];
  }
  return 0;
}
// track elements (global keypoint indices) of the ADD_POINT operations of the log, concatenated in log order
int64_t mpsfm_triangulator_num_op_elements(mpsfm_triangulator* h) { return h ? (int64_t)h->op_els.size() : MPSFM_EINVAL; }
int mpsfm_triangulator_get_op_elements(mpsfm_triangulator* h, int64_t* els) {
  if (int rc = check_engine(h)) return rc;
  std::copy(h->op_els.begin(), h->op_els.end(), els);
  return 0;
}
// EstimateTriangulation of independent candidate tracks in one k_tri_ransac launch (the engine's batch, exposed for callers
// that hold candidate sets themselves and for the parity tests against oracle/track_graph_oracle.py).
int mpsfm_tri_estimate_batch(const mpsfm_tri_candidates* c, int32_t device, double* xyz, uint8_t* ok, uint8_t* inlier) {
  if (!c || c->n_candidates < 0 || (c->n_candidates > 0 && (!c->cand_start || !xyz || !ok))) return gfail(MPSFM_EINVAL, "NULL argument");
  const int64_t nc = c->n_candidates;
  const int64_t nv = nc > 0 ? c->cand_start[nc] : 0;
  if (nv > 0 && (!c->view_cam_from_world || !c->view_intr || !c->view_xy || !inlier)) return gfail(MPSFM_EINVAL, "view arrays are NULL");
  if (c->residual_type != TRI_RESIDUAL_ANGULAR && c->residual_type != TRI_RESIDUAL_REPROJECTION) return gfail(MPSFM_EINVAL, "unknown residual type");
  for (int64_t i = 0; i < nc; ++i) {
    const int64_t n = c->cand_start[i + 1] - c->cand_start[i];
    if (n < 0) return gfail(MPSFM_EINVAL, "cand_start must be non-decreasing");
    if (n > kTriMaxViews) return gfail(MPSFM_EUNSUPPORTED, "a candidate track exceeds 64 views (the batch kernel's scratch)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gfail(MPSFM_ENODEVICE, "no HIP device visible: libmpsfm_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return gfail(MPSFM_EINVAL, "device ordinal out of range");
  if (device >= kMaxDevices) return gfail(MPSFM_EUNSUPPORTED, "device ordinals beyond 15 are not supported (per-device pools)");
  if (nc == 0) return 0;
  std::vector<TriCand> cands((size_t)nc);
  std::vector<TriView> views((size_t)nv);
  for (int64_t v = 0; v < nv; ++v) {
    const double* P = c->view_cam_from_world + 12 * v;
    const double R[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    const double t[3] = {P[3], P[7], P[11]};
    tri_make_view(R, t, c->view_intr + 4 * v, c->view_xy + 2 * v, views[(size_t)v]);
  }
  for (int64_t i = 0; i < nc; ++i) {
    const int64_t n = c->cand_start[i + 1] - c->cand_start[i];
    TriRansacOptions r{};
    r.min_tri_angle = c->min_tri_angle; r.max_error = c->max_error; r.confidence = 0.9999; r.max_num_trials = 10000;
    r.min_num_trials = c->min_num_trials ? c->min_num_trials[i] : (n <= 15 ? n * (n - 1) / 2 : 0);
    r.residual_type = c->residual_type;
    cands[(size_t)i].v0 = c->cand_start[i]; cands[(size_t)i].n = (int32_t)n; cands[(size_t)i].opt = r;
  }
  if (hipSetDevice(device) != hipSuccess) return gfail(MPSFM_EHIP, "hipSetDevice failed");
  std::vector<TriResult> results((size_t)nc);
  TriCand* d_c = (TriCand*)cached_malloc(sizeof(TriCand) * (size_t)nc);
  TriView* d_v = (TriView*)cached_malloc(sizeof(TriView) * (size_t)(nv > 0 ? nv : 1));
  TriResult* d_r = (TriResult*)cached_malloc(sizeof(TriResult) * (size_t)nc);
  int rc = 0;
  if (!d_c || !d_v || !d_r) rc = gfail(MPSFM_ENOMEM, "hipMalloc failed");
  if (!rc) rc = staged_upload(d_c, cands.data(), sizeof(TriCand) * (size_t)nc);
  if (!rc && nv > 0) rc = staged_upload(d_v, views.data(), sizeof(TriView) * (size_t)nv);
  if (!rc) rc = staged_drain();
  hipStream_t st = nullptr;
  if (!rc && pooled_stream(&st) != hipSuccess) rc = gfail(MPSFM_EHIP, "hipStreamCreate failed");
  if (!rc) {
    hipLaunchKernelGGL(k_tri_ransac, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, st, d_c, d_v, (int)nc, d_r);
    if (hipMemcpyAsync(results.data(), d_r, sizeof(TriResult) * (size_t)nc, hipMemcpyDeviceToHost, st) != hipSuccess) rc = gfail(MPSFM_EHIP, "reading the RANSAC batch back failed");
  }
  if (st) { (void)hipStreamSynchronize(st); release_stream(st); }
  cached_free(d_c); cached_free(d_v); cached_free(d_r);
  if (rc) return rc;
  for (int64_t i = 0; i < nc; ++i) {
    const TriResult& r = results[(size_t)i];
    ok[i] = (uint8_t)r.ok;
    for (int k = 0; k < 3; ++k) xyz[3 * i + k] = r.ok ? r.X[k] : 0.0;
    for (int64_t v = c->cand_start[i]; v < c->cand_start[i + 1]; ++v) inlier[v] = r.ok ? (uint8_t)(r.mask >> (v - c->cand_start[i]) & 1) : 0;
  }
  return 0;
}

int mpsfm_triangulator_stats(mpsfm_triangulator* h, int64_t* batch, int64_t* batch_hits, int64_t* host_estimates) {
  if (int rc = check_engine(h)) return rc;
  if (batch) *batch = h->n_batch;
  if (batch_hits) *batch_hits = h->n_batch_hits;
  if (host_estimates) *host_estimates = h->n_host_estimates;
  return 0;
}

}  // extern "C"
