// Host side of the level-scheduled tile Cholesky of the reduced camera system (dense_chol.hip: k_chol_level).
//
// Ceres factors the reduced camera matrix with a sparse Cholesky after a fill-reducing ordering (reference
// mpsfm/sfm/mapper/bundle_adjustment.py:288, SPARSE_SCHUR).  Here the factorisation runs on 32 x 32 tiles, one launch per
// LEVEL of the tile elimination tree: a right-looking step costs ~9 us whatever it holds (launch boundary, cold operand
// load, one 32-column factorisation in a wave), so what counts is the length of the dependent chain.  In the caller's
// camera order an orbit / video sequence is a band: one chain of nt steps.  A nested-dissection order (separators from
// breadth-first level structures of the camera graph, arcs in Cuthill-McKee order) turns it into independent chains that
// advance in the same launches: C3 38 -> 14 levels, C4 188 -> 20, for ~20 % more tile products.
//
//   camera graph -> order_cameras() -> slots and their first columns (padding columns — identity rows of the system — end
//   every non-root segment on a 32-column tile boundary, so no tile straddles two segments) -> tile pattern -> symbolic
//   factorisation, elimination tree, levels -> item tables.
#pragma once
#include <cstdint>
#include <vector>

namespace mpsfm {

struct CholItem {  // one workgroup of a level launch
  uint16_t type, ti, tk, nsrc;  // panel / trail: tile (ti, tk) and the number of source columns; role: (j, k, number of rows, -)
  uint32_t src;                 // offset of the source list (panel, trail) in srcs
  uint32_t aux;                 // role: offset of the row list in rows
};
static_assert(sizeof(CholItem) == 16, "CholItem is read as one 16-byte scalar load");
constexpr uint16_t kItemPanel = 0, kItemTrail = 1, kItemRole = 2;
constexpr int32_t kSrcX = 1 << 16;  // panel source flag: the source column also reaches the item's own tile (ti in struct(src))

struct CamGraph {  // symmetric adjacency of the variable cameras in the caller's order, one bit row per camera
  int n = 0, words = 0;
  std::vector<uint64_t> bits;
  void init(int n_) { n = n_; words = (n_ + 63) / 64; bits.assign((size_t)n * (size_t)words, 0); }
  void set(int a, int b) { bits[(size_t)a * words + (b >> 6)] |= 1ull << (b & 63); }
  bool get(int a, int b) const { return (bits[(size_t)a * words + (b >> 6)] >> (b & 63)) & 1; }
  const uint64_t* row(int a) const { return bits.data() + (size_t)a * words; }
};

struct CholPlan {
  int ncv = 0, nslots = 0, n = 0, nt = 0, nlevels = 0;
  int nd_depth = -1;  // -1: caller's order
  bool use_pinv = false;
  std::vector<int32_t> slot_of_nat;  // [ncv]    variable camera (caller's order) -> slot (a permutation: nslots == ncv)
  std::vector<int32_t> nat_of_slot;  // [nslots] slot -> variable camera
  std::vector<int32_t> col_of_slot;  // [nslots] first of the slot's six columns in the reduced system (n columns in all)
  std::vector<int32_t> slot_of_col;  // [n]      slot * 8 + coordinate of every column, -1: padding (identity row)
  std::vector<int32_t> struct_start, struct_rows;  // per tile column j: the rows i > j with L(i,j) != 0, ascending, the rhs row nt last
  std::vector<int32_t> parent, level;              // tile elimination tree
  std::vector<CholItem> items;
  std::vector<int32_t> launch_start;  // [nlevels + 1] item range of every launch
  std::vector<int32_t> srcs, rows;
  std::vector<int32_t> asm_tiles;     // packed ids (lt_tile) of the tiles the factorisation touches
  std::vector<int32_t> back_cols, back_start;  // backward substitution: columns by level, highest level first
  int64_t products = 0, roles = 0;
  double est_us = 0.0;
};

// Tile pattern: nt x nt bytes, pat[ti * nt + tj] != 0 for ti > tj: tile (ti, tj) of S can be nonzero.
// tables = false: symbolic factorisation, levels and the cost estimate only (plan_auto compares candidates that way)
void plan_from_pattern(const std::vector<uint8_t>& pat, int nt, bool use_pinv, int inv_rows, CholPlan& P, bool tables = true);
// slot order from the camera graph.  depth < 0: caller's order; depth >= 0: nested dissection of that depth (0: components + RCM)
// move_up: a segment that exceeds a whole number of tiles by at most this many cameras hands them to its parent's separator
void order_cameras(const CamGraph& g, int depth, int move_up, std::vector<int32_t>& slot_of_nat, std::vector<int32_t>& col_of_slot, int& n_cols);
void tile_pattern(const CamGraph& g, const std::vector<int32_t>& slot_of_nat, const std::vector<int32_t>& col_of_slot, int n_cols,
                  std::vector<uint8_t>& pat, int& nt);
// tries the caller's order and dissection depths 0 .. max, keeps the cheapest by the launch-cost model (forced_depth >= -1: that one)
// pfor(n, fn, ctx): runs fn(ctx, i) for i in [0, n), possibly on several threads (the caller's worker pool); NULL: one after the other
typedef void (*PlanParallelFor)(int n, void (*fn)(void* ctx, int i), void* ctx);
void plan_auto(const CamGraph& g, int forced_depth, bool forced, int pinv_max_tiles, int inv_rows, CholPlan& P, PlanParallelFor pfor = nullptr);

}  // namespace mpsfm
