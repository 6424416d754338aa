// The decisions of one Levenberg-Marquardt iteration (Ceres' TrustRegionMinimizer, trust_region_minimizer.cc, in its order): used
// by the decision kernels of the launch chain (ba_kernels.hip) and by the single-launch solver of small problems (local_lm.hip),
// where every workgroup takes the same decisions from the same scalars and only one of them (C != NULL) writes the traces.
#pragma once
#include "common.h"
#include <cfloat>

namespace mpsfm {

// The head of the block and the iteration's scalars are pulled into registers first and the head is written back once: as a
// chain of dependent global loads and stores the same logic took ~15 us.
__device__ __forceinline__ void lm_decide_logic(LmHead& L, LmCtl* C, const double (&sc)[U_COUNT], const LmOpts& o) {
  auto trace = [&](double cost, double rad, int acc) {
    if (L.trace_len < MPSFM_MAX_TRACE) {
      if (C) { C->trace_cost[L.trace_len] = cost; C->trace_radius[L.trace_len] = rad; C->trace_accepted[L.trace_len] = (uint8_t)acc; }
      L.trace_len++;
    }
  };
  auto next = [&]() {  // the tests at the top of the next iteration
    if (L.term != kLmRunning) return;
    if (L.iter >= o.max_iterations) L.term = MPSFM_TERM_MAX_ITERATIONS;
    else if (L.radius <= o.min_radius) L.term = MPSFM_TERM_MIN_RADIUS;
  };
  L.accepted = 0;
  L.iter += 1; L.n_jac_evals += 1; L.n_cost_evals += 1;
  const int chol_fail = sc[U_CHOL_FAIL] != 0.0 ? 1 : 0;
  L.last_chol_fail = chol_fail;
  const double x_cost = sc[U_X_COST];
  const bool x_bad = sc[U_X_BAD] > 0.0;  // residual not evaluable or a landmark block not positive definite
  L.last_x_cost = x_cost;
  if (L.iter == 1) {
    if (!isfinite(x_cost) || x_bad) { L.term = kLmNumericError; return; }
    L.initial_cost = x_cost + L.fixed_cost;
    L.cur_cost = x_cost;
    trace(x_cost + L.fixed_cost, L.radius, 1);
  }
  if (L.check_gradient) {  // Ceres checks the gradient tolerance at iteration 0 and after each successful step
    L.check_gradient = 0;
    const double gmax = fmax(sc[U_GMAX_CAMS], sc[U_GMAX_PTS]);
    if (gmax <= o.gradient_tolerance) { L.term = MPSFM_TERM_GRADIENT_TOLERANCE; L.iter -= 1; L.n_cost_evals -= 1; return; }
  }
  if (L.iter == 1) {  // the iteration and radius limits are looked at after iteration 0 (cost and gradient at the start) was evaluated
    if (o.max_iterations <= 0) { L.term = MPSFM_TERM_MAX_ITERATIONS; L.iter = 0; L.n_cost_evals -= 1; return; }
    if (L.radius <= o.min_radius) { L.term = MPSFM_TERM_MIN_RADIUS; L.iter = 0; L.n_cost_evals -= 1; return; }
  }
  const double mcc = sc[U_MCC];
  L.last_mcc = mcc;
  const bool solver_ok = !x_bad && chol_fail == 0 && isfinite(mcc);
  if (!(solver_ok && mcc > 0.0)) {
    L.invalid_run += 1; L.n_unsuccess += 1;
    if (L.invalid_run >= o.max_invalid_steps) L.term = MPSFM_TERM_INVALID_STEPS;
    L.radius /= L.decrease_factor; L.decrease_factor *= 2.0;
    trace(L.cur_cost + L.fixed_cost, L.radius, 0);
    L.last_cand = DBL_MAX; L.last_rel = 0.0; L.last_step_norm = 0.0;
    next();
    return;
  }
  L.invalid_run = 0;
  const double cand = (sc[U_BAD] > 0.0 || !isfinite(sc[U_CAND_COST])) ? DBL_MAX : sc[U_CAND_COST];
  const double step_norm = sqrt(sc[U_STEP_SQ_PTS] + sc[U_STEP_SQ_CAMS]);
  L.last_cand = cand; L.last_step_norm = step_norm;
  if (step_norm <= o.parameter_tolerance * (L.x_norm + o.parameter_tolerance)) { L.term = MPSFM_TERM_PARAMETER_TOLERANCE; return; }
  const double cost_change = x_cost - cand;
  if (fabs(cost_change) <= o.function_tolerance * x_cost) { L.term = MPSFM_TERM_FUNCTION_TOLERANCE; return; }
  const double rel = cost_change / mcc;
  L.last_rel = rel;
  if (rel > o.min_relative_decrease) {
    L.accepted = 1;
    L.x_norm = sqrt(sc[U_XN_SQ_PTS] + sc[U_XN_SQ_CAMS]);
    L.cur_cost = cand;
    const double u = 2.0 * rel - 1.0;
    L.radius = fmin(o.max_radius, L.radius / fmax(1.0 / 3.0, 1.0 - u * u * u));
    L.decrease_factor = 2.0;
    L.n_success += 1;
    L.check_gradient = 1;
    trace(cand + L.fixed_cost, L.radius, 1);
  } else {
    L.radius /= L.decrease_factor; L.decrease_factor *= 2.0;
    L.n_unsuccess += 1;
    trace(L.cur_cost + L.fixed_cost, L.radius, 0);
  }
  next();
}
}  // namespace mpsfm
