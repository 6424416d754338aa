"""triangulate_image / complete_and_merge on the NumPy scene (30 cameras / 20 k landmarks): wall time per call and the Python side."""
import cProfile, pstats, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from test_gpu_triangulator import empty_scene, OPTS, MpsfmTriangulator
sc, cg, prob, truth = empty_scene(30, 20000, 5, false_matches=2000, outlier_frac=0.02)
tri = MpsfmTriangulator({"colmap_options": dict(OPTS), "lift_low_parallax": False}, sc, cg)
tri._require_engine()
ts = []
ids = sorted(sc.images)
pr = cProfile.Profile()
for k, imid in enumerate(ids):
    sc.images[imid].has_pose = True
    if k == 20:
        pr.enable()
    t0 = time.perf_counter(); n = tri._triangulator.triangulate_image(tri.options, imid); ts.append(1e3 * (time.perf_counter() - t0))
pr.disable()
print("triangulate_image wall ms per image:", [round(t, 1) for t in ts])
print("keypoints per image ~", int(np.mean([len(im.kps) for im in sc.images.values()])), "points", len(sc.points3D))
t0 = time.perf_counter(); n = tri.complete_and_merge_all_tracks(); print("complete_and_merge_all_tracks %.1f ms (%d)" % (1e3 * (time.perf_counter() - t0), n))
t0 = time.perf_counter(); n = tri.retriangulate(); print("retriangulate %.1f ms (%d)" % (1e3 * (time.perf_counter() - t0), n))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
