"""One-off extended fuzz of small problems (mostly the single-launch path) against the oracle."""
import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
from oracle import cpu_oracle as O
L = capi.lib()
L.mpsfm_debug_local_clocks.argtypes = [C.c_void_p, C.c_void_p]; L.mpsfm_debug_local_clocks.restype = C.c_int
def problem(seed):
    rng = np.random.default_rng(5000 + seed)
    n_cams = int(rng.integers(2, 20)); n_pts = int(rng.integers(20, 6000)); depth = bool(rng.integers(0, 2))
    prob, _ = make_scene(n_cams, n_pts, depth, seed=seed, outlier_frac=float(rng.choice([0.0, 0.05, 0.2])), max_track=int(rng.choice([3, 8, 20])))
    if rng.random() < 0.5: prob.pt_const[rng.random(prob.n_pts) < 0.15] = 1
    if rng.random() < 0.5 and n_cams > 3: prob.pose_const[rng.choice(np.arange(1, n_cams), size=max(1, n_cams // 4), replace=False)] = 1
    if rng.random() < 0.3: prob.reproj_loss_type = int(rng.integers(0, 3))
    return prob
n_local = bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    pg, po = problem(seed), problem(seed)
    try:
        so = O.solve(po)
    except Exception as e:
        try:
            capi.ba_solve(pg); print(seed, "oracle failed, GPU did not:", e); bad += 1
        except Exception:
            pass
        continue
    with capi.BAHandle(pg) as h:
        sg = h.solve(); h.get_state(pg)
        clk = (C.c_int64 * 12)(); loc = L.mpsfm_debug_local_clocks(h._h, clk)
    n_local += int(loc)
    ok = (sg["termination"] == so["termination"] and abs(sg["num_iterations"] - so["num_iterations"]) <= 1
          and abs(sg["final_cost"] - so["final_cost"]) <= 1e-6 * abs(so["final_cost"]) and abs(sg["initial_cost"] - so["initial_cost"]) <= 1e-11 * abs(so["initial_cost"]))
    if ok and sg["num_iterations"] == so["num_iterations"] and sg["termination"] != "max_iterations":
        ok = np.allclose(pg.cam_t, po.cam_t, atol=1e-5)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "local" if loc else "chain", sg["termination"], so["termination"], sg["num_iterations"], so["num_iterations"], sg["final_cost"], so["final_cost"], flush=True)
print("seeds", sys.argv[1], sys.argv[2], "single-launch solves", n_local, "mismatches", bad)
