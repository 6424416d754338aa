// Device-side table build (build_dev.hip), called from build() in ba_solver.hip.
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

namespace mpsfm {

constexpr int MPSFM_DEVBUILD_FALLBACK = 1;  // stage2: the problem needs the host build (long tracks); nothing was produced

struct DevBuildOut {
  // device tables, owned by the receiver (cached_malloc blocks)
  ChunkHdr* d_chunks = nullptr;
  int32_t *d_chunk_cams = nullptr, *d_rec_cam = nullptr, *d_rec_pt = nullptr, *d_pt_rec_start = nullptr, *d_fx_cam = nullptr, *d_fx_pt = nullptr;
  uint32_t *d_rec_meta = nullptr, *d_fx_meta = nullptr;
  uint16_t* d_pt_kv = nullptr;
  double *d_rec_xy = nullptr, *d_rec_d = nullptr, *d_rec_m = nullptr, *d_rec_a = nullptr, *d_fx_xy = nullptr, *d_fx_d = nullptr, *d_fx_m = nullptr, *d_fx_a = nullptr;
  // host copies of the small tables the rest of build() works on
  std::vector<ChunkHdr> chunks;
  std::vector<int32_t> chunk_cams, order;
  int64_t np = 0, np_chunked = 0, n_long = 0, nrec = 0, nfixed = 0, nblk_reduced = 0;
  double nvarpts = 0;
  void release();
};

class DevBuilder {
 public:
  DevBuilder();
  ~DevBuilder();
  DevBuilder(const DevBuilder&) = delete;
  DevBuilder& operator=(const DevBuilder&) = delete;
  // uploads the observation lists, groups the blocks by landmark, returns the blocks per camera, the camera graph over the caller's
  // ("natural") slots and the longest block list of a landmark
  int stage1(const mpsfm_ba_problem* P, hipStream_t stream, const std::vector<int32_t>& nat_slot, int ncv_real, std::vector<double>& cam_counts,
             std::vector<uint64_t>& graph_bits, int graph_words, int64_t* max_blocks_per_landmark);
  // with the final camera slots: landmark order, chunk cut, record arrays.  Returns 0, MPSFM_DEVBUILD_FALLBACK or an error code.
  // rec_cap: records a DENSE chunk may hold (kObsMax, or less for small problems); pts_by_cams: dense_pts_cap by the camera set (build())
  int stage2(const std::vector<int32_t>& slot_of_cam, bool dense_on, int rec_cap, int pts_by_cams, DevBuildOut& out);
  // the reduction tables of the dense chunks' slabs (k_reduce_slabs) from device copies of the chunk headers (slab offsets set) and
  // camera lists; d_diag_block[slot]: the block of S on that slot's diagonal or -1.  The two tables are the receiver's
  // (cached_malloc blocks); stage1 must have run (the builder's stream).
  int slab_tables(const ChunkHdr* d_chunks, int n_dense, const int32_t* d_chunk_cams, const BlockSky& sky, int64_t nsb, int ncv, const int32_t* d_diag_block,
                  RedDest** d_dests, int32_t* n_dests, int32_t** d_srcs, int64_t* n_srcs);

 private:
  struct Impl;
  Impl* m;
};

}  // namespace mpsfm
