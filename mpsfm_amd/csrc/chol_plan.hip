// Ordering, symbolic factorisation and launch tables of the level-scheduled tile Cholesky (host code only; see chol_plan.h).
#include "chol_plan.h"
#include <thread>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>

#include "../../include/mpsfm_hip.h"

namespace mpsfm {
namespace {

constexpr int kTileCols = 32;    // every segment but the last ends on a tile boundary (padding columns)
constexpr int kMinLeaf = 32;     // no dissection below this many cameras
constexpr int kMinBfsLevels = 5;
constexpr double kMaxPlanProducts = 3.0e6;  // tile products beyond which no item table is built (tens of MB of tables, seconds to build)

using Mask = std::vector<uint64_t>;

struct Sub {  // breadth-first search inside a node subset
  const CamGraph& g;
  Mask in, seen;
  explicit Sub(const CamGraph& g_) : g(g_), in((size_t)g_.words, 0), seen((size_t)g_.words, 0) {}
  void set_nodes(const std::vector<int>& nodes) {
    std::fill(in.begin(), in.end(), 0);
    for (int v : nodes) in[(size_t)(v >> 6)] |= 1ull << (v & 63);
  }
  int degree(int v) const {
    int d = 0;
    const uint64_t* r = g.row(v);
    for (int w = 0; w < g.words; ++w) d += __builtin_popcountll(r[w] & in[(size_t)w]);
    return d;
  }
  // order: nodes by level, level_start: offsets (last entry = size).  Only nodes of `in` that are not in `seen` yet.
  void bfs(int start, std::vector<int>& order, std::vector<int>& level_start, bool reset_seen) {
    if (reset_seen) std::fill(seen.begin(), seen.end(), 0);
    order.clear(); level_start.clear();
    order.push_back(start); seen[(size_t)(start >> 6)] |= 1ull << (start & 63);
    level_start.push_back(0);
    size_t lo = 0;
    while (lo < order.size()) {
      const size_t hi = order.size();
      for (size_t q = lo; q < hi; ++q) {
        const uint64_t* r = g.row(order[q]);
        for (int w = 0; w < g.words; ++w) {
          uint64_t m = r[w] & in[(size_t)w] & ~seen[(size_t)w];
          seen[(size_t)w] |= m;
          while (m) { order.push_back(w * 64 + __builtin_ctzll(m)); m &= m - 1; }
        }
      }
      lo = hi;
      if (order.size() > hi) level_start.push_back((int)hi);
    }
    level_start.push_back((int)order.size());
  }
};

void components(Sub& S, const std::vector<int>& nodes, std::vector<std::vector<int>>& comps) {
  comps.clear();
  S.set_nodes(nodes);
  std::fill(S.seen.begin(), S.seen.end(), 0);
  std::vector<int> order, ls;
  for (int v : nodes) {
    if ((S.seen[(size_t)(v >> 6)] >> (v & 63)) & 1) continue;
    S.bfs(v, order, ls, false);
    comps.push_back(order);
    std::sort(comps.back().begin(), comps.back().end());
  }
}

// a node far from the "middle" of a connected subset: repeated BFS from a lowest-degree node of the last level
int pseudo_peripheral(Sub& S, const std::vector<int>& nodes, std::vector<int>& order, std::vector<int>& ls) {
  S.set_nodes(nodes);
  int s = nodes[0], best = 1 << 30;
  for (int v : nodes) { const int d = S.degree(v); if (d < best) { best = d; s = v; } }
  int depth = -1;
  for (int it = 0; it < 8; ++it) {
    S.bfs(s, order, ls, true);
    const int nl = (int)ls.size() - 1;
    if (nl <= depth) break;
    depth = nl;
    int cand = s, bd = 1 << 30;
    for (int q = ls[(size_t)nl - 1]; q < ls[(size_t)nl]; ++q) { const int d = S.degree(order[(size_t)q]); if (d < bd) { bd = d; cand = order[(size_t)q]; } }
    if (cand == s) break;
    s = cand;
  }
  S.bfs(s, order, ls, true);
  return s;
}

// reverse Cuthill-McKee of a subset (every component from a pseudo-peripheral node)
void rcm(Sub& S, const std::vector<int>& nodes, std::vector<int>& out) {
  std::vector<std::vector<int>> comps;
  components(S, nodes, comps);
  std::vector<int> order, ls;
  for (const auto& c : comps) {
    pseudo_peripheral(S, c, order, ls);
    out.insert(out.end(), order.rbegin(), order.rend());
  }
}

struct Seg { std::vector<int> cams; std::vector<int> child; };

int dissect(Sub& S, std::vector<Seg>& tree, const std::vector<int>& nodes, int depth) {
  const int me = (int)tree.size();
  tree.emplace_back();
  std::vector<std::vector<int>> comps;
  components(S, nodes, comps);
  if (comps.size() > 1) {  // independent parts: an empty separator above them
    for (const auto& c : comps) { const int ch = dissect(S, tree, c, depth); tree[(size_t)me].child.push_back(ch); }
    return me;
  }
  std::vector<int> order, ls;
  if (depth > 0 && (int)nodes.size() >= kMinLeaf) {
    pseudo_peripheral(S, nodes, order, ls);
    const int nl = (int)ls.size() - 1;
    if (nl >= kMinBfsLevels) {
      // the level whose removal leaves the smallest larger side plus itself
      int bm = -1; long best = 1L << 60;
      for (int m = 1; m < nl - 1; ++m) {
        const long a = ls[(size_t)m], b = (long)order.size() - ls[(size_t)m + 1], s = ls[(size_t)m + 1] - ls[(size_t)m];
        const long cost = std::max(a, b) + s;
        if (cost < best) { best = cost; bm = m; }
      }
      if (bm > 0) {
        std::vector<int> A(order.begin(), order.begin() + ls[(size_t)bm]), B(order.begin() + ls[(size_t)bm + 1], order.end()),
            sep(order.begin() + ls[(size_t)bm], order.begin() + ls[(size_t)bm + 1]);
        // cameras of the level that see nothing beyond it are not needed in the separator: a level is as wide as the LONGEST
        // link out of the level before it, most cameras reach less far
        {
          S.set_nodes(B);
          std::vector<int> keep;
          for (int v : sep) { if (S.degree(v) > 0) keep.push_back(v); else A.push_back(v); }
          if (!keep.empty()) sep.swap(keep);
          else { for (size_t q = 0; q < sep.size(); ++q) A.pop_back(); }  // (a level without forward links cannot be a middle level)
        }
        std::sort(A.begin(), A.end()); std::sort(B.begin(), B.end()); std::sort(sep.begin(), sep.end());
        const int ca = dissect(S, tree, A, depth - 1);
        const int cb = dissect(S, tree, B, depth - 1);
        tree[(size_t)me].child = {ca, cb};
        rcm(S, sep, tree[(size_t)me].cams);
        return me;
      }
    }
  }
  rcm(S, nodes, tree[(size_t)me].cams);
  return me;
}

// Post-order flattening.  kSegEnd after every segment but the last: the next camera starts on a tile boundary (padding
// columns in between).  A segment that exceeds a whole number of tiles by a few cameras hands them to its parent instead
// (a separator stays one when it grows), so that its chain is a tile shorter.
constexpr int kSegEnd = -1;
void flatten(const std::vector<Seg>& tree, int node, bool is_root, int move_up, std::vector<int>& out, std::vector<int>& up) {
  std::vector<int> cams;
  for (int c : tree[(size_t)node].child) {
    std::vector<int> moved;
    flatten(tree, c, false, move_up, out, moved);
    cams.insert(cams.end(), moved.begin(), moved.end());
  }
  const std::vector<int>& own = tree[(size_t)node].cams;
  if (!is_root && own.empty()) { up = cams; return; }  // independent parts under an empty separator: what moved up keeps moving
  cams.insert(cams.end(), own.begin(), own.end());
  if (cams.empty()) return;
  if (!is_root) {
    const int k = (int)cams.size();
    const int whole = (6 * k) / kTileCols * kTileCols / 6;  // cameras that fit the whole tiles of the segment
    const int over = k - whole;
    if (over > 0 && over <= move_up && whole > 0 && (6 * k) % kTileCols != 0) {
      up.assign(cams.end() - over, cams.end());
      cams.resize((size_t)whole);
    }
  }
  out.insert(out.end(), cams.begin(), cams.end());
  if (!is_root) out.push_back(kSegEnd);
}

}  // namespace

void order_cameras(const CamGraph& g, int depth, int move_up, std::vector<int32_t>& slot_of_nat, std::vector<int32_t>& col_of_slot, int& n_cols) {
  slot_of_nat.assign((size_t)g.n, -1);
  col_of_slot.assign((size_t)g.n, 0);
  if (depth < 0 || g.n == 0) {
    for (int i = 0; i < g.n; ++i) { slot_of_nat[(size_t)i] = i; col_of_slot[(size_t)i] = 6 * i; }
    n_cols = 6 * g.n;
    return;
  }
  Sub S(g);
  std::vector<Seg> tree;
  std::vector<int> all((size_t)g.n);
  for (int i = 0; i < g.n; ++i) all[(size_t)i] = i;
  dissect(S, tree, all, depth);
  std::vector<int> out, up;
  flatten(tree, 0, true, move_up, out, up);
  while (!out.empty() && out.back() == kSegEnd) out.pop_back();  // nothing follows the last segment
  int slot = 0, col = 0;
  for (int v : out) {
    if (v == kSegEnd) { col = (col + kTileCols - 1) / kTileCols * kTileCols; continue; }
    slot_of_nat[(size_t)v] = slot;
    col_of_slot[(size_t)slot] = col;
    ++slot; col += 6;
  }
  n_cols = col;
}

void tile_pattern(const CamGraph& g, const std::vector<int32_t>& slot_of_nat, const std::vector<int32_t>& col_of_slot, int n_cols,
                  std::vector<uint8_t>& pat, int& nt) {
  nt = (n_cols + 31) / 32;
  pat.assign((size_t)nt * (size_t)nt, 0);
  auto mark = [&](int sa, int sb) {
    const int ca = col_of_slot[(size_t)sa], cb = col_of_slot[(size_t)sb];
    const int a0 = ca / 32, a1 = (ca + 5) / 32, b0 = cb / 32, b1 = (cb + 5) / 32;
    for (int a = a0; a <= a1; ++a)
      for (int b = b0; b <= b1; ++b) { if (a > b) pat[(size_t)a * nt + b] = 1; else if (b > a) pat[(size_t)b * nt + a] = 1; }
  };
  for (int i = 0; i < g.n; ++i) {
    const int si = slot_of_nat[(size_t)i];
    mark(si, si);
    const uint64_t* r = g.row(i);
    for (int w = 0; w < g.words; ++w) {
      uint64_t m = r[w];
      while (m) {
        const int j = w * 64 + __builtin_ctzll(m);
        m &= m - 1;
        if (j > i && j < g.n) mark(si, slot_of_nat[(size_t)j]);
      }
    }
  }
}

void plan_from_pattern(const std::vector<uint8_t>& pat, int nt, bool use_pinv, int inv_rows, CholPlan& P, bool tables) {
  P.nt = nt; P.use_pinv = use_pinv;
  P.struct_start.assign((size_t)nt + 1, 0); P.struct_rows.clear();
  P.parent.assign((size_t)nt, -1); P.level.assign((size_t)nt, 0);
  P.items.clear(); P.launch_start.clear(); P.srcs.clear(); P.rows.clear(); P.asm_tiles.clear();
  P.back_cols.clear(); P.back_start.clear();
  P.products = 0; P.roles = 0; P.nlevels = 0;
  if (nt <= 0) { P.launch_start.push_back(0); P.back_start.push_back(0); return; }
  const int W = (nt + 1 + 63) / 64;
  // symbolic factorisation: struct(j) = pattern of column j below the diagonal, plus struct(c) \ {j} of the children c
  std::vector<uint64_t> st((size_t)nt * (size_t)W, 0);
  std::vector<std::vector<int>> children((size_t)nt);
  auto bit = [&](int j, int i) -> bool { return (st[(size_t)j * W + (i >> 6)] >> (i & 63)) & 1; };
  for (int j = 0; j < nt; ++j) {
    uint64_t* s = st.data() + (size_t)j * W;
    for (int i = j + 1; i < nt; ++i) if (pat[(size_t)i * nt + j]) s[i >> 6] |= 1ull << (i & 63);
    s[nt >> 6] |= 1ull << (nt & 63);  // the right-hand-side row
    for (int c : children[(size_t)j]) {
      const uint64_t* sc = st.data() + (size_t)c * W;
      for (int w = 0; w < W; ++w) s[w] |= sc[w];
    }
    for (int i = 0; i <= j; ++i) s[i >> 6] &= ~(1ull << (i & 63));
    int par = -1;
    for (int i = j + 1; i < nt; ++i) if (bit(j, i)) { par = i; break; }
    P.parent[(size_t)j] = par;
    if (par >= 0) children[(size_t)par].push_back(j);
    int lv = 0;
    for (int c : children[(size_t)j]) lv = std::max(lv, P.level[(size_t)c] + 1);
    P.level[(size_t)j] = lv;
    P.nlevels = std::max(P.nlevels, lv + 1);
  }
  for (int j = 0; j < nt; ++j) {
    for (int i = j + 1; i <= nt; ++i) if (bit(j, i)) P.struct_rows.push_back(i);
    P.struct_start[(size_t)j + 1] = (int32_t)P.struct_rows.size();
  }
  // a reduced system without exploitable structure would need a table of ~nt^3/6 tile products: leave the item tables
  // empty (nlevels = 0) and let the caller fall back to the dense outer-panel path
  {
    double prod = 0.0;
    for (int j = 0; j < nt; ++j) { const double sj = (double)(P.struct_start[(size_t)j + 1] - P.struct_start[(size_t)j]); prod += 0.5 * sj * (sj + 1.0); }
    if (prod > kMaxPlanProducts) {
      P.products = (int64_t)prod; P.nlevels = 0; P.est_us = 1e30;
      P.launch_start.assign(1, 0); P.back_start.assign(1, 0);
      return;
    }
    if (!tables) {  // candidate comparison: every column's struct(j) (struct(j) + 1) / 2 tile products, roles over the subtree sizes
      P.products = (int64_t)prod;
      if (use_pinv) {
        std::vector<int64_t> sub((size_t)nt, 1);
        for (int j = 0; j < nt; ++j) {
          if (P.parent[(size_t)j] >= 0) sub[(size_t)P.parent[(size_t)j]] += sub[(size_t)j];
          const int64_t nrr = P.struct_start[(size_t)j + 1] - P.struct_start[(size_t)j] - 1;
          P.roles += sub[(size_t)j] * ((nrr + inv_rows - 1) / std::max(inv_rows, 1));
        }
      }
      P.est_us = 9.5 * P.nlevels + 0.02 * (double)P.products + 0.004 * (double)P.roles + (use_pinv ? 10.0 : 6.5 * P.nlevels);
      return;
    }
  }
  auto rows_of = [&](int j) { return std::make_pair(P.struct_rows.data() + P.struct_start[(size_t)j], P.struct_rows.data() + P.struct_start[(size_t)j + 1]); };
  auto lt = [](int64_t ti, int64_t tj) { return (int32_t)(ti * (ti + 1) / 2 + tj); };
  for (int j = 0; j < nt; ++j) {
    P.asm_tiles.push_back(lt(j, j));
    auto r = rows_of(j);
    for (const int32_t* p = r.first; p != r.second; ++p) P.asm_tiles.push_back(lt(*p, j));
  }
  std::sort(P.asm_tiles.begin(), P.asm_tiles.end());
  // subtree lists (inverse roles): the descendants of j, j included
  std::vector<std::vector<int>> subtree;
  if (use_pinv) {
    subtree.resize((size_t)nt);
    for (int j = 0; j < nt; ++j) {
      for (int c : children[(size_t)j]) subtree[(size_t)j].insert(subtree[(size_t)j].end(), subtree[(size_t)c].begin(), subtree[(size_t)c].end());
      subtree[(size_t)j].push_back(j);
      std::sort(subtree[(size_t)j].begin(), subtree[(size_t)j].end());
    }
  }
  std::vector<std::vector<int>> by_level((size_t)P.nlevels);
  for (int j = 0; j < nt; ++j) by_level[(size_t)P.level[(size_t)j]].push_back(j);
  // launch l: the panels of the columns of level l; trailing updates and inverse roles from the columns of level l-1
  for (int l = 0; l < P.nlevels; ++l) {
    P.launch_start.push_back((int32_t)P.items.size());
    for (int c : by_level[(size_t)l]) {
      // children whose level is l-1 reach column c through the panel items; deeper children went through trailing items
      std::vector<int> kids;
      for (int k : children[(size_t)c]) if (P.level[(size_t)k] == l - 1) kids.push_back(k);
      auto r = rows_of(c);
      for (int q = -1; q < (int)(r.second - r.first); ++q) {
        const int ti = q < 0 ? c : r.first[q];
        CholItem it{kItemPanel, (uint16_t)ti, (uint16_t)c, (uint16_t)kids.size(), (uint32_t)P.srcs.size(), 0};
        for (int k : kids) P.srcs.push_back(k | ((ti != c && bit(k, ti)) ? kSrcX : 0));
        P.items.push_back(it);
      }
    }
    if (l == 0) continue;
    if (use_pinv) {
      for (int j : by_level[(size_t)l - 1]) {
        auto r = rows_of(j);
        const int nrr = (int)(r.second - r.first) - 1;  // without the right-hand-side row
        for (int g0 = 0; g0 < nrr; g0 += inv_rows) {
          const int nr = std::min(inv_rows, nrr - g0);
          const uint32_t off = (uint32_t)P.rows.size();
          for (int q = 0; q < nr; ++q) P.rows.push_back(r.first[g0 + q]);
          for (int k : subtree[(size_t)j]) { P.items.push_back(CholItem{kItemRole, (uint16_t)j, (uint16_t)k, (uint16_t)nr, 0, off}); ++P.roles; }
        }
      }
    }
    std::map<std::pair<int, int>, std::vector<int>> trail;  // (tk, ti) -> source columns
    for (int k : by_level[(size_t)l - 1]) {
      auto r = rows_of(k);
      const int par = P.parent[(size_t)k];
      for (const int32_t* pk = r.first; pk != r.second; ++pk) {
        const int tk = *pk;
        if (tk >= nt) continue;
        if (tk == par && P.level[(size_t)par] == l) continue;  // the panel items of column par take this source
        for (const int32_t* pi = pk; pi != r.second; ++pi) trail[{tk, *pi}].push_back(k);
      }
    }
    for (auto& e : trail) {
      CholItem it{kItemTrail, (uint16_t)e.first.second, (uint16_t)e.first.first, (uint16_t)e.second.size(), (uint32_t)P.srcs.size(), 0};
      for (int k : e.second) P.srcs.push_back(k);
      P.items.push_back(it);
      P.products += (int64_t)e.second.size();
    }
  }
  P.launch_start.push_back((int32_t)P.items.size());
  for (const CholItem& it : P.items) if (it.type == kItemPanel) P.products += it.nsrc;
  for (int l = P.nlevels - 1; l >= 0; --l) {
    P.back_start.push_back((int32_t)P.back_cols.size());
    for (int j : by_level[(size_t)l]) P.back_cols.push_back(j);
  }
  P.back_start.push_back((int32_t)P.back_cols.size());
  // launch-cost model (measured on MI355X: a dependent launch ~9.5 us whatever it holds up to a few hundred workgroups)
  P.est_us = 9.5 * P.nlevels + 0.02 * (double)P.products + 0.004 * (double)P.roles + (use_pinv ? 10.0 : 6.5 * P.nlevels);
}

void plan_auto(const CamGraph& g, int forced_depth, bool forced, int pinv_max_tiles, int inv_rows, CholPlan& best, PlanParallelFor pfor) {
  int dmax = -1;
  for (int s = g.n; s >= kMinLeaf / 2; s /= 2) ++dmax;  // candidate depths down to parts of ~kMinLeaf / 2 cameras (dissect() itself
  dmax = std::min(dmax, 6);                             // stops at kMinLeaf); the cost model decides (C3: depth 3, 14 levels)
  // the candidate orders are independent: evaluated side by side on host threads when the graph is large (C4: 6 ms one after
  // the other), the cheapest by the launch-cost model wins, ties go to the first in this list (what the sequential loop chose)
  struct Cand { int depth, move_up; CholPlan P; std::vector<uint8_t> pat; };
  std::vector<Cand> cands;
  for (int d = -1; d <= (forced ? -1 : dmax); ++d) {
    const int depth = forced ? forced_depth : d;
    for (int move_up : {0, 2, 4}) {
      if (depth < 1 && move_up > 0) break;  // nothing to move without separators
      cands.push_back(Cand{depth, move_up, CholPlan(), {}});
    }
  }
  auto evaluate = [&](Cand& c) {
    CholPlan& P = c.P;
    P.ncv = g.n; P.nd_depth = c.depth;
    P.nslots = g.n;
    order_cameras(g, c.depth, c.move_up, P.slot_of_nat, P.col_of_slot, P.n);
    int nt = 0;
    tile_pattern(g, P.slot_of_nat, P.col_of_slot, P.n, c.pat, nt);
    plan_from_pattern(c.pat, nt, nt <= pinv_max_tiles, inv_rows, P, /*tables=*/false);
  };
  if (pfor && g.n >= 64 && cands.size() > 1) {  // the caller's persistent workers (C3: 0.5 -> 0.15 ms; C4: 6.3 -> 1.9 ms)
    struct Ctx { decltype(evaluate)* ev; std::vector<Cand>* c; } ctx{&evaluate, &cands};
    pfor((int)cands.size(), [](void* p, int i) { Ctx* x = static_cast<Ctx*>(p); (*x->ev)((*x->c)[(size_t)i]); }, &ctx);
  } else if (g.n >= 256 && cands.size() > 1) {
    std::vector<std::thread> th;
    for (size_t i = 1; i < cands.size(); ++i) th.emplace_back([&, i] { evaluate(cands[i]); });
    evaluate(cands[0]);
    for (auto& x : th) x.join();
  } else {
    for (Cand& c : cands) evaluate(c);
  }
  size_t bi = 0;
  for (size_t i = 1; i < cands.size(); ++i)
    if (cands[i].P.est_us < cands[bi].P.est_us - 1e-9) bi = i;
  best = std::move(cands[bi].P);
  std::vector<uint8_t> best_pat = std::move(cands[bi].pat);
  plan_from_pattern(best_pat, best.nt, best.nt <= pinv_max_tiles, inv_rows, best, /*tables=*/true);
  best.nat_of_slot.assign((size_t)best.nslots, -1);
  for (int i = 0; i < g.n; ++i) best.nat_of_slot[(size_t)best.slot_of_nat[(size_t)i]] = i;
  best.slot_of_col.assign((size_t)best.n, -1);
  for (int sl = 0; sl < best.nslots; ++sl)
    for (int a = 0; a < 6; ++a) best.slot_of_col[(size_t)best.col_of_slot[(size_t)sl] + a] = sl * 8 + a;
}

}  // namespace mpsfm

// ---- test hook: the plan of a camera graph, without a device (tests/test_chol_plan_cpu.py interprets the item tables with
// NumPy and compares with a dense Cholesky) ------------------------------------------------------------------------------
extern "C" {

struct mpsfm_plan_handle { mpsfm::CholPlan P; };

// adj: n x n bytes (symmetric, nonzero = the cameras share a landmark).  depth -2: choose; >= -1: that order.
mpsfm_plan_handle* mpsfm_debug_plan_create(const uint8_t* adj, int32_t n, int32_t depth, int32_t pinv_max_tiles, int32_t inv_rows) {
  if (!adj || n < 0) return nullptr;
  mpsfm::CamGraph g;
  g.init(n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) if (i != j && adj[(size_t)i * n + j]) g.set(i, j);
  auto* h = new mpsfm_plan_handle();
  mpsfm::plan_auto(g, depth, depth >= -1, pinv_max_tiles, inv_rows, h->P);
  return h;
}
void mpsfm_debug_plan_destroy(mpsfm_plan_handle* h) { delete h; }
// what: 0 header {ncv, nslots, n, nt, nlevels, nd_depth, use_pinv, n_items, products, roles}; 1 slot_of_nat; 2 struct_start; 3 struct_rows;
// 4 parent; 5 level; 6 items (4 int32 each: type | ti << 16, tk | nsrc << 16, src, aux); 7 launch_start; 8 srcs; 9 rows; 10 asm_tiles;
// 11 back_cols; 12 back_start; 13 col_of_slot.  Returns the length; copies min(length, cap) entries.
int64_t mpsfm_debug_plan_get(const mpsfm_plan_handle* h, int32_t what, int32_t* out, int64_t cap) {
  if (!h) return -1;
  const mpsfm::CholPlan& P = h->P;
  std::vector<int32_t> tmp;
  const std::vector<int32_t>* v = &tmp;
  switch (what) {
    case 0: tmp = {P.ncv, P.nslots, P.n, P.nt, P.nlevels, P.nd_depth, P.use_pinv ? 1 : 0, (int32_t)P.items.size(), (int32_t)P.products, (int32_t)P.roles}; break;
    case 1: v = &P.slot_of_nat; break;
    case 2: v = &P.struct_start; break;
    case 3: v = &P.struct_rows; break;
    case 4: v = &P.parent; break;
    case 5: v = &P.level; break;
    case 6:
      for (const mpsfm::CholItem& it : P.items) {
        tmp.push_back((int32_t)(it.type | ((uint32_t)it.ti << 16))); tmp.push_back((int32_t)(it.tk | ((uint32_t)it.nsrc << 16)));
        tmp.push_back((int32_t)it.src); tmp.push_back((int32_t)it.aux);
      }
      break;
    case 7: v = &P.launch_start; break;
    case 8: v = &P.srcs; break;
    case 9: v = &P.rows; break;
    case 10: v = &P.asm_tiles; break;
    case 11: v = &P.back_cols; break;
    case 12: v = &P.back_start; break;
    case 13: v = &P.col_of_slot; break;
    default: return -1;
  }
  if (out) std::memcpy(out, v->data(), sizeof(int32_t) * (size_t)std::min<int64_t>(cap, (int64_t)v->size()));
  return (int64_t)v->size();
}

}  // extern "C"
