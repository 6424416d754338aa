"""CPU checks of oracle/track_graph_oracle.py (row f2's independent restatement of COLMAP 3.11's EstimateTriangulation and
IncrementalTriangulator): known answers of the estimator, the stop rule, and the graph walk on a synthetic scene.  The HIP
engine is compared with this oracle in tests/test_gpu_triangulator.py."""

import math

import numpy as np

from mpsfm_amd.sfm.mapper.track_engine import graph_arrays, state_arrays
from mpsfm_amd.synthetic import make_scene
from numpy_scene import correspondences_from_problem, scene_from_problem
from oracle import track_graph_oracle as TG

K = np.array([1200.0, 1190.0, 800.0, 600.0])


def _views(X, n, rng, noise=0.3, outliers=()):
    views = []
    for i in range(n):
        ang = 0.25 * i
        C = np.array([6 * math.sin(ang), 0.3 * rng.normal(), -6 * math.cos(ang)])
        z = -C / np.linalg.norm(C)
        x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
        R = np.stack([x, np.cross(z, x), z])
        t = -R @ C
        P = np.hstack([R, t[:, None]])
        pc = R @ X + t
        xy = np.array([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]]) + rng.normal(0, noise, 2)
        if i in outliers:
            xy += np.array([150.0, -90.0])
        views.append(TG.View(xy=xy, xn=np.array([(xy[0] - K[2]) / K[0], (xy[1] - K[3]) / K[1]]), P=P, C=C, K=K))
    return views


def test_estimator_known_answers():
    rng = np.random.default_rng(0)
    X = np.array([0.3, -0.2, 0.5])
    o = TG.RansacOptions(max_error=math.radians(2.0), min_tri_angle=math.radians(1.5), min_num_trials=10)
    rep = TG.loransac_estimate(_views(X, 5, rng), o)
    assert rep.success and rep.inlier_mask.all() and np.linalg.norm(rep.model - X) < 5e-3
    assert rep.num_trials == 10  # exhaustive: C(5,2) samples, no early stop below min_num_trials
    rep = TG.loransac_estimate(_views(X, 6, rng, outliers=(2,)), TG.RansacOptions(max_error=math.radians(2.0), min_num_trials=15))
    assert rep.success and list(rep.inlier_mask) == [True, True, False, True, True, True] and np.linalg.norm(rep.model - X) < 5e-3
    # reprojection residual in pixels: same decision on this track
    rep = TG.loransac_estimate(_views(X, 6, rng, outliers=(4,)), TG.RansacOptions(max_error=4.0, residual_type=TG.REPROJECTION_ERROR, min_num_trials=15))
    assert rep.success and list(rep.inlier_mask) == [True, True, True, True, False, True]
    # two views that see the point under less than the minimum angle: no model
    v = _views(X, 2, rng)
    v[1] = TG.View(xy=v[0].xy + 0.2, xn=v[0].xn + 0.2 / K[0], P=v[0].P.copy(), C=v[0].C + 1e-4, K=K)
    v[1].P[:, 3] = -v[1].P[:, :3] @ v[1].C
    assert not TG.loransac_estimate(v, TG.RansacOptions(max_error=math.radians(2.0), min_tri_angle=math.radians(1.5), min_num_trials=1)).success
    # a point behind both cameras (rays diverge): cheirality rejects
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = np.hstack([np.eye(3), np.array([[-1.0], [0], [0]])])   # second camera at x = +1, both looking along +z
    v = [TG.View(xy=np.zeros(2), xn=np.array([-0.1, 0.0]), P=P1, C=np.zeros(3), K=K),
         TG.View(xy=np.zeros(2), xn=np.array([0.1, 0.0]), P=P2, C=np.array([1.0, 0, 0]), K=K)]
    assert np.allclose(TG.triangulate_point(P1, P2, v[0].xn, v[1].xn), [0.5, 0, -5.0])
    assert not TG.loransac_estimate(v, TG.RansacOptions(max_error=math.radians(2.0), min_num_trials=1)).success
    assert not TG.loransac_estimate(_views(X, 1, rng), o).success


def test_stop_rule_and_geometry_helpers():
    assert TG.compute_num_trials(10, 10, 0.9999, 3.0) == 1
    assert TG.compute_num_trials(0, 10, 0.9999, 3.0) == 2**63 - 1
    assert TG.compute_num_trials(5, 10, 0.9999, 3.0) == math.ceil(math.log(1e-4) / math.log(0.75) * 3.0)
    # 30 clean views: the first sample's local optimisation takes every view, the dynamic bound drops to 1 and the walk stops
    # after the second trial (min_num_trials = 0 beyond 15 views)
    rng = np.random.default_rng(1)
    rep = TG.loransac_estimate(_views(np.array([0.1, 0.1, 0.2]), 30, rng, noise=0.1), TG.RansacOptions(max_error=math.radians(2.0)))
    assert rep.success and rep.inlier_mask.all() and rep.num_trials == 3
    C1, C2, X = np.array([0, 0, 0.0]), np.array([1, 0, 0.0]), np.array([0.5, 0, 0.5])
    assert TG.calculate_triangulation_angle(C1, C2, X) == np.pi / 2
    assert TG.calculate_triangulation_angle(C1, C1, X) == 0.0
    P = np.hstack([np.eye(3), np.zeros((3, 1))])
    assert TG.calculate_squared_reprojection_error(np.array([800.0, 600.0]), np.array([0, 0, -1.0]), P, K) == np.finfo(float).max
    assert TG.calculate_squared_reprojection_error(np.array([803.0, 604.0]), np.array([0, 0, 2.0]), P, K) == 25.0
    assert abs(TG.calculate_normalized_angular_error(np.array([0.0, 0.0]), np.array([1.0, 0, 1.0]), P) - np.pi / 4) < 1e-15


def _scene(n_cams, n_pts, seed, false_matches):
    prob, truth = make_scene(n_cams, n_pts, True, seed=seed, perturb=False, outlier_frac=0.02)
    sc = scene_from_problem(prob, truth, seed=seed, with_points=False)
    cg = correspondences_from_problem(sc, prob, false_matches=false_matches, seed=seed)
    return sc, cg, prob, truth


def test_graph_walk_recovers_a_scene_and_keeps_its_books():
    sc, cg, prob, truth = _scene(7, 250, 3, 40)
    ga = graph_arrays(cg, sc)
    st, _ = state_arrays(sc, ga["image_ids"], ga["kp_start"])
    o = TG.TrackGraphOracle(ga["kp_start"], ga["kp_xy"], ga["intr"], ga["corr_start"], ga["corr_kp"])
    opts = {"min_angle": 0.001, "ignore_two_view_tracks": False}
    reg = np.zeros(len(ga["image_ids"]), bool)
    o.set_state(reg, st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
    total = 0
    for i in range(len(reg)):
        o.registered[i] = True
        total += o.triangulate_image(opts, i)
    assert total == sum(len(p.elements) for p in o.points.values()) and len(o.points) > 0.8 * prob.n_pts
    # books: a keypoint belongs to at most one point, both directions agree, only registered images
    seen = set()
    for pid, p in o.points.items():
        assert len(p.elements) >= 2
        for kp in p.elements:
            assert kp not in seen and o.kp_point[kp] == pid
            seen.add(kp)
    assert (o.kp_point >= 0).sum() == len(seen)
    # positions: through the first element's landmark
    obs_of = {}
    for oi, (i, k) in enumerate(zip(sc._obs_image, sc._obs_point2D)):
        obs_of[int(ga["kp_start"][ga["im_index"][int(i)]] + k)] = oi
    err = [np.linalg.norm(p.xyz - truth["pts"][prob.obs_pt[obs_of[p.elements[0]]]]) for p in o.points.values()]
    assert np.median(err) < 0.05
    # Complete is idempotent on a complete scene; Merge never leaves a keypoint in two points; Retriangulate spends one trial
    n1 = o.complete_tracks(opts)
    assert o.complete_tracks(opts) == 0 and n1 >= 0
    o.merge_tracks(opts)
    assert len(set(k for p in o.points.values() for k in p.elements)) == sum(len(p.elements) for p in o.points.values())
    o.retriangulate(dict(opts, re_min_ratio=1.1))
    trials = dict(o.re_num_trials)
    assert trials and all(v == 1 for v in trials.values())
    assert o.retriangulate(dict(opts, re_min_ratio=1.1)) == 0 and o.re_num_trials == trials and o.ops == []
