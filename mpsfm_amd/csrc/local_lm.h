// Single-launch Levenberg-Marquardt for small problems (local bundle adjustment): interface between ba_solver.hip and local_lm.hip.
#pragma once
#include "common.h"

namespace mpsfm {

constexpr int kLocalCams = 16;            // variable cameras the single-launch solver takes (three 32-column tiles)
constexpr int kLocalN = 6 * kLocalCams;   // reduced dimension, padded
// one dense accumulator of the reduced system: S (upper block triangle, row-major kLocalN x kLocalN) | g_c | W V^-1 g_p | diag U
constexpr int kLocalAccDoubles = kLocalN * kLocalN + 3 * kLocalN;

struct LocalArgs {
  SweepArgs A;            // tables and state of the handle (ctl = NULL: the radius travels in the workgroups' own control block)
  LmCtl* ctl;             // in: the initial control block; out: the final one with its traces
  LmOpts o;
  LmHead* log;            // [max_iterations + 2] control-block heads, one per iteration (verbose runs), or NULL
  double* acc[2];         // two zeroed accumulators of kLocalAccDoubles doubles (iterations alternate)
  int32_t* bar;           // [0] arrivals of the grid barrier (monotonic), [1] abort flag; zeroed by the host
  long long* clk;         // [7] wall-clock ticks (100 MHz): sweep + flush, barrier 1, dense + cameras, update sweep, barrier 2, decision; iterations
  int32_t ncv, nc, nchunks, pad_;
  double* q; double* t;   // [nc][4], [nc][3] camera state, written back at the end
  double* camtab;         // [nc][kCamRec] likewise (A.camtab is its read-only view)
  double* pts;            // [np][3] landmark state (A.pts is its read-only view)
  const double* cs;       // [nc][6] camera column scales
  const double* fixed_parts;  // [2] cost of the fixed blocks (reprojection, depth): summed into the control block at the start
};

// co-resident workgroups the device offers the kernel (0: cooperative launches unavailable)
int local_lm_max_chunks(int device);
// enqueues the solve; returns a hipError_t as int
int launch_local_lm(const LocalArgs& a, hipStream_t s);

}  // namespace mpsfm
