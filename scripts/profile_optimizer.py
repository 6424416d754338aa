"""cProfile of one Optimizer.ba() call on the NumPy scene (40 cameras / 30 k landmarks): where the Python side spends its time."""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from mpsfm_amd.synthetic import make_scene
from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from numpy_scene import scene_from_problem
prob, truth = make_scene(40, 30000, True, seed=5)
sc = scene_from_problem(prob, truth, seed=1)
og = Optimizer({}, sc, None)
b = {"optim_ids": set(sc.images), "pts3D": set(sc.points3D), "constpoints": set()}
for _ in range(2):
    og.ba(b, mode="global", allow_scale_filter=True)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); og.ba(b, mode="global", allow_scale_filter=True); ts.append(1e3 * (time.perf_counter() - t0))
print("wall ms", [round(t, 1) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    og.ba(b, mode="global", allow_scale_filter=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
