"""Row f2: the native track-graph engine (csrc/triangulator.hip) behind MpsfmTriangulator(engine=None).  The fork's
IncrementalTriangulator is not available (parity unpinned): what is tested are the semantics it restates from COLMAP
3.11 — invariants of every operation, the GPU batch against the same arithmetic on the host, recovery of a synthetic
scene, and an incremental run with bundle adjustment and retriangulation (BASELINE config 5 in miniature)."""

import copy

import numpy as np
import pytest

from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from mpsfm_amd.sfm.mapper.triangulator import MpsfmTriangulator, track_quality
from numpy_scene import INVALID_POINT3D, correspondences_from_problem, scene_from_problem
from mpsfm_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu

OPTS = {"min_angle": 0.001, "ignore_two_view_tracks": False}  # the mapper's overrides (reference mapper/base.py:35-40)


def empty_scene(n_cams, n_pts, seed, false_matches=0, **kw):
    prob, truth = make_scene(n_cams, n_pts, True, seed=seed, perturb=False, **kw)
    sc = scene_from_problem(prob, truth, seed=seed, with_points=False)
    cg = correspondences_from_problem(sc, prob, false_matches=false_matches, seed=seed)
    for im in sc.images.values():
        im.has_pose = False
    return sc, cg, prob, truth


def check_consistency(sc):
    """every keypoint belongs to at most one track and the two directions of the bookkeeping agree"""
    seen = set()
    for pid, p in sc.points3D.items():
        assert p.track.length() >= 2
        for e in p.track.elements:
            assert (e.image_id, e.point2D_idx) not in seen
            seen.add((e.image_id, e.point2D_idx))
            assert sc.images[e.image_id].kp_point3D[e.point2D_idx] == pid
            assert sc.images[e.image_id].has_pose
    n_assigned = sum(int((im.kp_point3D != INVALID_POINT3D).sum()) for im in sc.images.values())
    assert n_assigned == len(seen)


def register_all(sc, tri):
    tri._require_engine()
    total = 0
    for imid in sorted(sc.images):
        sc.images[imid].has_pose = True
        total += tri._triangulator.triangulate_image(tri.options, imid)
    return total


def test_incremental_triangulation_recovers_the_scene():
    sc, cg, prob, truth = empty_scene(10, 1500, 71, false_matches=300, outlier_frac=0.02)
    tri = MpsfmTriangulator({"colmap_options": dict(OPTS), "lift_low_parallax": False}, sc, cg)
    n = register_all(sc, tri)
    check_consistency(sc)
    assert n == sum(p.track.length() for p in sc.points3D.values())  # counts = observations that joined a point
    assert len(sc.points3D) > 0.85 * prob.n_pts
    # every point sits where its landmark is: match through the first track element
    first_obs = {(int(i), int(k)): o for o, (i, k) in enumerate(zip(sc._obs_image, sc._obs_point2D))}
    err = []
    for p in sc.points3D.values():
        e = p.track.elements[0]
        err.append(np.linalg.norm(p.xyz - truth["pts"][prob.obs_pt[first_obs[(e.image_id, e.point2D_idx)]]]))
    assert np.median(err) < 0.05 and np.mean(np.array(err) < 0.5) > 0.97
    # the estimator's acceptance rules hold for what was created: angular error of every element <= 2 degrees,
    # reprojection of every element in front of its camera
    ids = sorted(sc.points3D)
    ang, sq_err, front = track_quality(sc, ids)
    assert front.all() and (ang >= np.deg2rad(OPTS["min_angle"])).mean() > 0.999
    st = tri._triangulator.stats()
    assert st["batch_candidates"] > 1000 and st["batch_hits"] > 0.5 * st["batch_candidates"]
    print("triangulated", len(sc.points3D), "points;", st)


def test_gpu_batch_equals_the_same_arithmetic_on_the_host(monkeypatch):
    results = []
    for host in ("0", "1"):
        monkeypatch.setenv("MPSFM_TRI_HOST_BATCH", host)
        sc, cg, prob, truth = empty_scene(8, 900, 73, false_matches=150)
        tri = MpsfmTriangulator({"colmap_options": dict(OPTS), "lift_low_parallax": False}, sc, cg)
        n = register_all(sc, tri)
        results.append((n, {pid: (tuple(np.round(p.xyz, 9)), tuple((e.image_id, e.point2D_idx) for e in p.track.elements)) for pid, p in sc.points3D.items()}))
    assert results[0][0] == results[1][0] and results[0][1] == results[1][1]


def test_complete_merge_retriangulate_semantics():
    sc, cg, prob, truth = empty_scene(9, 1200, 75)
    tri = MpsfmTriangulator({"colmap_options": dict(OPTS), "lift_low_parallax": False, "new_retry_nbatch": None}, sc, cg)
    register_all(sc, tri)
    check_consistency(sc)
    # --- complete: observations taken out of their tracks come back (their reprojection error is small)
    rng = np.random.default_rng(0)
    long_pts = [pid for pid, p in sc.points3D.items() if p.track.length() >= 4]
    removed = []
    for pid in rng.choice(long_pts, 150, replace=False):
        p = sc.points3D[int(pid)]
        e = p.track.elements[-1]
        im = sc.images[e.image_id]
        Xc = (im.cam_from_world * p.xyz[None])[0]
        K = sc.rec.cameras[im.camera_id].params
        err2 = (K[0] * Xc[0] / Xc[2] + K[2] - im.kps[e.point2D_idx][0]) ** 2 + (K[1] * Xc[1] / Xc[2] + K[3] - im.kps[e.point2D_idx][1]) ** 2
        removed.append((int(pid), e.image_id, e.point2D_idx, err2))
        sc.obs.delete_observation(e.image_id, e.point2D_idx)
    before = sum(p.track.length() for p in sc.points3D.values())
    n_c = tri.complete_all_tracks()
    after = sum(p.track.length() for p in sc.points3D.values())
    assert n_c == after - before
    # the rule: an observation comes back exactly when its squared reprojection error is within complete_max_reproj_error^2
    for pid, image_id, idx, err2 in removed:
        if abs(err2 - 16.0) > 1e-6:
            assert (sc.images[image_id].kp_point3D[idx] == pid) == (err2 <= 16.0) or sc.images[image_id].kp_point3D[idx] not in (pid, INVALID_POINT3D)
    assert n_c >= sum(1 for r in removed if r[3] <= 16.0) > 50
    check_consistency(sc)
    assert tri.complete_all_tracks() == 0  # nothing left to complete
    # --- merge: a track split into two points is joined again; the merged point replaces both
    split = [pid for pid, p in sc.points3D.items() if p.track.length() >= 6][:60]
    for pid in split:
        p = sc.points3D[pid]
        els = p.track.elements[3:]
        for e in els:
            sc.obs.delete_observation(e.image_id, e.point2D_idx)
        tr = sc.Track()
        for e in els:
            tr.add_element(e.image_id, e.point2D_idx)
        sc.obs.add_point3D(p.xyz + rng.normal(0, 1e-3, 3), tr)
    n_before, id_before = len(sc.points3D), max(sc.points3D)
    n_m = tri.merge_all_tracks()
    merged = [pid for pid in sc.points3D if pid > id_before]
    assert n_m > 0 and len(merged) >= 10 and len(sc.points3D) == n_before - len(merged)  # two points became one, each time
    # the rule: a merge is accepted only when EVERY element of both tracks is within merge_max_reproj_error of the
    # length-weighted mean point
    _, sq_err, front = track_quality(sc, merged)
    assert front.all() and sq_err.max() <= 16.0 + 1e-9
    assert n_m == sum(sc.points3D[p].track.length() for p in merged)
    check_consistency(sc)
    assert tri.merge_all_tracks() == 0  # idempotent
    n_cm = tri.complete_and_merge_tracks(set(list(sc.points3D)[:50]))
    assert isinstance(n_cm, int) and n_cm >= 0
    # --- retriangulate: an image pair whose common points are gone is under-reconstructed and gets its points back
    ids = sorted(sc.images)
    a, b = ids[2], ids[3]
    gone = [pid for pid, p in sc.points3D.items() if {a, b} <= {e.image_id for e in p.track.elements}]
    assert len(gone) > 20
    for pid in gone:
        sc.obs.delete_point3D(pid)
    eng = tri._triangulator
    n_ignored = eng.retriangulate(tri.options, {a})  # the fork's ignore_image_ids: pairs of an ignored image are skipped
    n_pts_mid = len(sc.points3D)
    n_r = tri.retriangulate()
    check_consistency(sc)
    assert n_r > 0 and len(sc.points3D) > n_pts_mid
    back = [p for p in sc.points3D.values() if {a, b} <= {e.image_id for e in p.track.elements}]
    assert len(back) >= 0.5 * len(gone)
    assert eng.retriangulate(tri.options, set()) == 0  # re_max_trials = 1: a pair is retried once
    print("complete", n_c, "merge", n_m, "retriangulate", n_r, "(with the pair's image ignored:", n_ignored, ")")


def test_incremental_mapping_with_ba_and_retriangulation():
    """BASELINE configuration 5 in miniature: images registered one by one with noisy poses, triangulate_image with
    depth lifting of low-parallax points, local refinement by the HIP solver, complete + merge, retriangulation."""
    prob, truth = make_scene(14, 2500, True, seed=79)            # perturbed poses = what PnP registration would give
    sc = scene_from_problem(prob, truth, seed=79, with_points=False)
    cg = correspondences_from_problem(sc, prob, false_matches=200, seed=1)
    for im in sc.images.values():
        im.has_pose = False
    tri = MpsfmTriangulator({"colmap_options": dict(OPTS)}, sc, cg)
    opt = Optimizer({}, sc, None)
    ids = sorted(sc.images)
    costs = []
    for k, imid in enumerate(ids):
        sc.images[imid].has_pose = True
        assert tri.triangulate_image(imid) is True
        if k >= 2 and len(sc.points3D) > 50:
            reg = [i for i in ids[: k + 1]]
            bundle = {"optim_ids": set(reg), "pts3D": set(sc.points3D), "constpoints": set()}
            r, _ = opt.ba(bundle, mode="global", allow_scale_filter=True)
            costs.append((r.summary["initial_cost"], r.summary["final_cost"]))
            sc.obs.filter_all_points3D(4.0, 0.001)
            tri.complete_and_merge_all_tracks()
    n_re = tri.retriangulate()
    check_consistency(sc)
    assert all(f <= i for i, f in costs) and len(sc.points3D) > 0.7 * prob.n_pts
    t_err = np.array([np.linalg.norm(sc.images[i].cam_from_world.translation - truth["cam_t"][i - ids[0]]) for i in ids[2:]])
    assert np.median(t_err) < 0.05  # poses pulled back to the truth (gauge: first camera fixed, scale by the second)
    print("points", len(sc.points3D), "of", prob.n_pts, "; retriangulated", n_re, "; median pose error", float(np.median(t_err)))
