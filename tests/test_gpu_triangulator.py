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


# ---- against the independent oracle (oracle/track_graph_oracle.py: NumPy SVD / eigh, upstream's function structure) ---------

def _candidates(rng, n_cands, residual_type):
    """Random candidate tracks as Create / CompleteImage would meet them: 2-24 views of one landmark with pixel noise, some
    views replaced by far-off measurements (outliers), some tracks made of unrelated points (false matches), some seen
    under a tiny baseline, some behind a camera."""
    from oracle import track_graph_oracle as TG

    K = np.array([1200.0, 1190.0, 800.0, 600.0])
    cand_start, Ps, Ks, xys, views_all = [0], [], [], [], []
    for c in range(n_cands):
        n = int(rng.choice([2, 2, 3, 3, 4, 5, 6, 8, 11, 15, 16, 24]))
        kind = rng.choice(["clean", "clean", "outliers", "outliers", "false", "narrow", "behind"])
        X = rng.uniform(-2, 2, 3)
        views = []
        for i in range(n):
            spread = 0.004 if kind == "narrow" else 1.0
            ang = rng.uniform(-1.0, 1.0) * spread
            C = np.array([8 * np.sin(ang), rng.uniform(-1, 1) * spread, -8 * np.cos(ang)])
            z = (np.zeros(3) - C) / np.linalg.norm(C)
            if kind == "behind" and i == 0:
                z = -z
            x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
            R = np.stack([x, np.cross(z, x), z])
            t = -R @ C
            Xi = rng.uniform(-2, 2, 3) if kind == "false" else X
            pc = R @ Xi + t
            xy = np.array([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]]) + rng.normal(0, 0.7, 2)
            if kind == "outliers" and rng.uniform() < 0.3:
                xy += rng.choice([-1, 1], 2) * rng.uniform(15, 200, 2)
            xy = xy.astype(np.float16).astype(np.float64) if abs(xy).max() < 6e4 else xy   # Point2D.xy is fp16-rounded
            P = np.hstack([R, t[:, None]])
            views.append(TG.View(xy=xy, xn=np.array([(xy[0] - K[2]) / K[0], (xy[1] - K[3]) / K[1]]), P=P, C=C, K=K))
            Ps.append(P.ravel()); Ks.append(K); xys.append(xy)
        views_all.append(views)
        cand_start.append(cand_start[-1] + n)
    return np.array(cand_start), np.array(Ps), np.array(Ks), np.array(xys), views_all


@pytest.mark.parametrize("residual_type", [0, 1])
def test_ransac_batch_kernel_equals_the_independent_oracle(residual_type):
    """k_tri_ransac (through mpsfm_tri_estimate_batch) vs oracle.track_graph_oracle.loransac_estimate on 1500 candidate
    tracks: success flags and inlier masks EXACT, points to 1e-9 — except candidates whose decision sits within 1e-7
    (relative) of a threshold in the oracle, which are only counted."""
    import math

    from mpsfm_amd import capi
    from oracle import track_graph_oracle as TG

    rng = np.random.default_rng(1234 + residual_type)
    cs, P, Kv, xy, views_all = _candidates(rng, 1500, residual_type)
    min_angle = math.radians(0.5)
    max_error = math.radians(2.0) if residual_type == 0 else 4.0
    X, ok, inl = capi.tri_estimate_batch(cs, P, Kv, xy, min_angle, max_error, residual_type)
    n_ok = n_marginal = 0
    worst = 0.0
    for c, views in enumerate(views_all):
        n = len(views)
        o = TG.RansacOptions(max_error=max_error, min_tri_angle=min_angle, residual_type=residual_type,
                             min_num_trials=n * (n - 1) // 2 if n <= 15 else 0)
        rep = TG.loransac_estimate(views, o)
        if rep.success and rep.margin < 1e-7:
            n_marginal += 1
            continue
        assert bool(ok[c]) == rep.success, (c, n, rep)
        if not rep.success:
            assert not inl[cs[c]:cs[c + 1]].any()
            continue
        n_ok += 1
        np.testing.assert_array_equal(inl[cs[c]:cs[c + 1]], rep.inlier_mask, err_msg=f"candidate {c}")
        d = np.abs(X[c] - rep.model).max() / (1.0 + np.abs(rep.model).max())
        worst = max(worst, d)
        cond_ok = rep.inlier_mask.sum() >= 2
        assert cond_ok and d < 1e-9, (c, n, d, X[c], rep.model)
    assert n_ok > 600 and n_marginal < 5 and (~ok).sum() > 100   # every outcome occurs
    print(f"residual type {residual_type}: {n_ok} estimated, {int((~ok).sum())} rejected, {n_marginal} marginal, worst point difference {worst:.2e}")


def _engine_ops(eng):
    o = eng.last_ops
    out, e0 = [], 0
    for k in range(len(o["type"])):
        if o["type"][k] == 0:
            out.append((0, int(o["a"][k]), tuple(int(x) for x in o["elements"][e0:e0 + o["b"][k]]), o["xyz"][k].copy()))
            e0 += int(o["b"][k])
        elif o["type"][k] == 1:
            out.append((1, int(o["a"][k]), int(o["b"][k])))
        else:
            out.append((2, int(o["a"][k])))
    return out


def _assert_same_ops(got, want, what):
    assert len(got) == len(want), (what, len(got), len(want))
    for k, (g, w) in enumerate(zip(got, want)):
        assert g[:3] == w[:3] if g[0] == 0 else g == w, (what, k, g, w)
        if g[0] == 0:
            np.testing.assert_allclose(g[3], w[3], rtol=0, atol=1e-9 * (1 + np.abs(w[3]).max()), err_msg=f"{what} op {k}")


def test_engine_operation_logs_equal_the_independent_oracle():
    """The engine's operation log (which keypoint joins which point, in which order, where the new points are) of
    triangulate_image for every image of a 10-camera scene, complete_tracks / merge_tracks (subset and all),
    complete_image and retriangulate with and without ignored images — EXACTLY the log of the oracle walking the same
    state; both option sets the reference's mapper uses (its overrides, COLMAP's defaults)."""
    from oracle import track_graph_oracle as TG

    for opts, seed in ((dict(OPTS), 81), ({}, 83)):
        sc, cg, prob, truth = empty_scene(10, 700, seed, false_matches=250, outlier_frac=0.03)
        tri = MpsfmTriangulator({"colmap_options": dict(opts), "lift_low_parallax": False}, sc, cg)
        tri._require_engine()
        eng = tri._triangulator
        orc = TG.TrackGraphOracle(eng.kp_start, eng.kp_xy, eng.intr, eng.corr_start, eng.corr_kp)
        o_opts = dict(tri.options) if isinstance(tri.options, dict) else {k: getattr(tri.options, k) for k in TG.DEFAULT_OPTIONS if hasattr(tri.options, k)}
        counts = dict(add_point=0, add_obs=0, delete=0)

        def both(name, *args, oracle_args=None):
            n_g = getattr(eng, name)(tri.options, *args)
            st = eng.last_state
            orc.set_state(st["registered"], st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
            n_o = getattr(orc, name)(o_opts, *(oracle_args if oracle_args is not None else args))
            _assert_same_ops(_engine_ops(eng), orc.ops, f"{name}{args}")
            assert n_g == n_o, (name, n_g, n_o)
            for op in orc.ops:
                counts[("add_point", "add_obs", "delete")[op[0]]] += 1
            check_consistency(sc)
            return n_g

        ids = sorted(sc.images)
        for imid in ids:
            sc.images[imid].has_pose = True
            both("triangulate_image", imid, oracle_args=(eng.im_index[imid],))
        assert len(sc.points3D) > 0.7 * prob.n_pts
        rng = np.random.default_rng(seed)
        # observations taken out of long tracks -> Complete has work; tracks split in two -> Merge has work
        long_pts = [pid for pid, p in sc.points3D.items() if p.track.length() >= 5]
        for pid in rng.choice(long_pts, 80, replace=False):
            e = sc.points3D[int(pid)].track.elements[-1]
            sc.obs.delete_observation(e.image_id, e.point2D_idx)
        for pid in [pid for pid, p in sc.points3D.items() if p.track.length() >= 6][:50]:
            p = sc.points3D[pid]
            els = p.track.elements[3:]
            for e in els:
                sc.obs.delete_observation(e.image_id, e.point2D_idx)
            tr = sc.Track()
            for e in els:
                tr.add_element(e.image_id, e.point2D_idx)
            sc.obs.add_point3D(p.xyz + rng.normal(0, 2e-3, 3), tr)

        def engine_ids(scene_ids):  # scene point ids -> the indices both sides use (position in the state's arrays)
            st, point_ids = state_arrays_of(eng)
            pos = {int(p): i for i, p in enumerate(point_ids)}
            return [pos[int(p)] for p in scene_ids if int(p) in pos]

        subset = sorted(sc.points3D)[::7]
        both("complete_tracks", subset, oracle_args=(engine_ids(subset),))
        subset = sorted(sc.points3D)[::5]
        both("merge_tracks", subset, oracle_args=(engine_ids(subset),))
        n = eng.complete_all_tracks(tri.options)
        st = eng.last_state
        orc.set_state(st["registered"], st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
        assert n == orc.complete_tracks(o_opts) and (_assert_same_ops(_engine_ops(eng), orc.ops, "complete_all") is None)
        n = eng.merge_all_tracks(tri.options)
        st = eng.last_state
        orc.set_state(st["registered"], st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
        assert n == orc.merge_tracks(o_opts) and (_assert_same_ops(_engine_ops(eng), orc.ops, "merge_all") is None)
        # points of two image pairs removed -> under-reconstructed pairs; CompleteImage creates from untriangulated keypoints,
        # Retriangulate (first with an ignored image, then without) brings the rest back
        a, b, c = ids[2], ids[3], ids[6]
        for pid in [pid for pid, p in sc.points3D.items() if {a, b} <= {e.image_id for e in p.track.elements} or c in {e.image_id for e in p.track.elements}]:
            sc.obs.delete_point3D(pid)
        both("complete_image", c, oracle_args=(eng.im_index[c],))
        re_opts = dict(o_opts, re_min_ratio=0.6)
        n_g = eng.retriangulate(dict(opts, re_min_ratio=0.6), {a})
        st = eng.last_state
        orc.set_state(st["registered"], st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
        assert n_g == orc.retriangulate(re_opts, {eng.im_index[a]})
        _assert_same_ops(_engine_ops(eng), orc.ops, "retriangulate(ignore)")
        n_g2 = eng.retriangulate(dict(opts, re_min_ratio=0.6), set())
        st = eng.last_state
        orc.set_state(st["registered"], st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
        assert n_g2 == orc.retriangulate(re_opts, set()) and n_g2 > 0
        _assert_same_ops(_engine_ops(eng), orc.ops, "retriangulate")
        assert {tuple(sorted(k)) for k in orc.re_num_trials} and orc.decision_margin > 1e-9
        check_consistency(sc)
        assert counts["add_point"] > 300 and counts["add_obs"] > 50 and counts["delete"] > 10, counts
        print("options", opts or "COLMAP defaults", counts, "smallest decision margin", orc.decision_margin)


def state_arrays_of(eng):
    from mpsfm_amd.sfm.mapper.track_engine import state_arrays

    return state_arrays(eng.rec, eng.image_ids, eng.kp_start)


def test_complete_image_uses_the_reprojection_residual_in_pixels():
    """CompleteImage estimates with REPROJECTION_ERROR / complete_max_reproj_error (pixels), Create with ANGULAR_ERROR /
    create_max_angle_error (degrees): a view 10 px off at f = 1200 is 0.48 degrees off — inside 2 degrees, outside 4 px — and
    with complete_max_reproj_error = 200 px a view 120 px (5.7 degrees) off is inside the pixel bound and outside the angular."""
    from mpsfm_amd import capi

    # one clean 6-view track with the last view displaced
    K = np.array([1200.0, 1200.0, 800.0, 600.0])
    X = np.array([0.2, -0.1, 0.3])
    Ps, xys = [], []
    for i in range(6):
        ang = -0.5 + 0.2 * i
        C = np.array([8 * np.sin(ang), 0.1 * i, -8 * np.cos(ang)])
        z = -C / np.linalg.norm(C)
        x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
        R = np.stack([x, np.cross(z, x), z])
        t = -R @ C
        pc = R @ X + t
        Ps.append(np.hstack([R, t[:, None]]).ravel())
        xys.append([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]])
    Ps, xys, Ks, cs = np.array(Ps), np.array(xys), np.tile(K, (6, 1)), np.array([0, 6])
    for off, ang_in, px_in, px_bound in ((10.0, True, False, 4.0), (120.0, False, True, 200.0)):
        x2 = xys.copy()
        x2[5, 0] += off
        _, ok_a, in_a = capi.tri_estimate_batch(cs, Ps, Ks, x2, 0.0, np.deg2rad(2.0), 0)
        _, ok_p, in_p = capi.tri_estimate_batch(cs, Ps, Ks, x2, 0.0, px_bound, 1)
        assert ok_a[0] and ok_p[0] and in_a[:5].all() and in_p[:5].all()
        assert bool(in_a[5]) == ang_in and bool(in_p[5]) == px_in, (off, in_a, in_p)


def test_candidate_tracks_beyond_the_batch_limit_are_estimated_on_the_host_like_the_oracle():
    """All 72 images registered before the first triangulate_image: every keypoint meets ~70 untriangulated correspondences,
    more than the 64 views a batch thread holds — those candidates are estimated on the host (same function, heap scratch)
    and the log still equals the oracle's; beyond 15 views the sampling is not exhaustive (stop rule decides)."""
    from oracle import track_graph_oracle as TG

    prob, truth = make_scene(72, 40, True, seed=91, perturb=False, outlier_frac=0.05, max_track=72, track_mean=90.0)
    sc = scene_from_problem(prob, truth, seed=91, with_points=False)
    cg = correspondences_from_problem(sc, prob, false_matches=30, seed=91)
    tri = MpsfmTriangulator({"colmap_options": dict(OPTS), "lift_low_parallax": False}, sc, cg)
    tri._require_engine()
    eng = tri._triangulator
    orc = TG.TrackGraphOracle(eng.kp_start, eng.kp_xy, eng.intr, eng.corr_start, eng.corr_kp)
    ids = sorted(sc.images)
    assert np.diff(eng.corr_start).max() > 64
    for imid in ids[:3]:
        n_g = eng.triangulate_image(tri.options, imid)
        st = eng.last_state
        orc.set_state(st["registered"], st["cam_quat_xyzw"], st["cam_t"], st["kp_point"], st["xyz"])
        assert n_g == orc.triangulate_image(dict(OPTS), eng.im_index[imid])
        _assert_same_ops(_engine_ops(eng), orc.ops, f"triangulate_image({imid})")
    assert max(p.track.length() for p in sc.points3D.values()) > 64 and eng.stats()["host_estimates"] > 10
    check_consistency(sc)
