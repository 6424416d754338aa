"""Optimizer.ba(mode="local") on the NumPy scene (40 cameras / 30 k landmarks, a window of six images): wall time of the call and
where the Python side spends it, beside the time inside mpsfm_ba_solve."""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from mpsfm_amd.synthetic import make_scene
from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from numpy_scene import scene_from_problem
prob, truth = make_scene(40, 30000, True, seed=5)
sc = scene_from_problem(prob, truth, seed=1)
og = Optimizer({}, sc, None)
ids = sorted(sc.images)
ref = ids[15]
pts = set(sc.images[ref].point3D_ids(sc.images[ref].get_observation_point2D_idxs()))
b = {"ref_id": ref, "optim_ids": set(ids[10:16]), "pts3D": pts, "constpoints": set()}
for _ in range(2):
    r, _ = og.ba(b, mode="local", allow_scale_filter=True)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); r, _ = og.ba(b, mode="local", allow_scale_filter=True); ts.append(1e3 * (time.perf_counter() - t0))
s = r.summary
print("Optimizer.ba(mode='local') wall ms", [round(t, 2) for t in ts], "| inside the library: %d iterations, %.2f ms solve" % (s["num_iterations"], 1e3 * s["time_total_s"]),
      "| residual blocks", s["num_residual_blocks"], "reduced dim", s["reduced_dim"])
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    og.ba(b, mode="local", allow_scale_filter=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
