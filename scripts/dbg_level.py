"""Level-scheduled factorisation against the caller's order on the same problem: dense-solve time, solution difference."""
import os, sys, numpy as np
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
depths = sys.argv[2:] or ["-1", "auto"]
prob, _ = make_config(cfg)
ref = None
for d in depths:
    if d == "auto": os.environ.pop("MPSFM_CHOL_ND", None)
    else: os.environ["MPSFM_CHOL_ND"] = d
    o = capi.default_options(); o.verbose = 2
    h = capi.BAHandle(prob, options=o)
    h.sweep_once(1e4)
    t = [h.dense_solve_once() for _ in range(10)][4:]
    y = h.dense_solution()
    if ref is None: ref = y
    print("ND=%s: dense solve %.3f ms (min %.3f), max|y - y_ref|/max|y| = %.2e" % (d, np.mean(t), np.min(t), np.abs(y - ref).max() / np.abs(ref).max()), flush=True)
    s = h.solve()
    print("   full solve: %d it, %.3f ms, dense %.3f sweep %.3f update %.3f, final cost %.12e" % (s["num_iterations"], 1e3 * s["time_total_s"], 1e3 * s["time_dense_s"], 1e3 * s["time_linearize_s"], 1e3 * s["time_update_s"], s["final_cost"]), flush=True)
    del h
