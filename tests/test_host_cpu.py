"""CPU tests of the host layer: prior sampling, robust statistics, scene accessors and the
Optimizer's problem assembly (checked by solving the assembled flat problem with the oracle)."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from mpsfm_amd.problem import LOSS_CAUCHY, LOSS_SOFT_L1, LOSS_TRIVIAL
from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from numpy_scene import scene_from_problem
from mpsfm_amd.sfm.scene.priorutils import bilinear_at_kps, fit_robust_gaussian_mad
from mpsfm_amd.synthetic import make_scene
from oracle import cpu_oracle as O


from backends import OracleBackend  # noqa: E402


def test_bilinear_sampling_matches_grid_sample_golden():
    z = np.load(os.path.join(GOLDEN, "grid_sample.npz"))
    for nm in ("data", "mask"):
        out = bilinear_at_kps(z[nm], z["kps"], float(z["sx"]), float(z["sy"]))
        np.testing.assert_allclose(out, z["sampled_" + nm], rtol=1e-12, atol=1e-12)


def test_mad_matches_golden():
    z = np.load(os.path.join(GOLDEN, "robust_stats.npz"))
    for i in range(3):
        mu, sigma = fit_robust_gaussian_mad(z[f"x{i}"])
        assert mu == pytest.approx(float(z[f"mu{i}"]), rel=1e-14)
        assert sigma == pytest.approx(float(z[f"sigma{i}"]), rel=1e-14)


@pytest.fixture(scope="module")
def scene():
    prob, truth = make_scene(8, 400, True, seed=21)
    return scene_from_problem(prob, truth, seed=3), prob, truth


def _global_bundle(sc):
    return {"optim_ids": set(sc.images.keys()), "pts3D": set(sc.points3D.keys()), "constpoints": set()}


def test_global_problem_assembly(scene):
    sc, prob, _ = scene
    opt = Optimizer({}, sc, None, backend=OracleBackend())
    flat, ss = opt._build_problem(_global_bundle(sc), fix_pose=False, fix_scale=True, mode="global", solve=False)
    p = flat.prob
    order = list(_global_bundle(sc)["optim_ids"])
    assert flat.image_ids == order
    assert p.n_obs == prob.n_obs and p.n_cams == 8
    assert list(p.pose_const) == [1] + [0] * 7 and p.gauge_axis_cam == 1
    assert p.pt_const.sum() == 0  # every track is fully inside a global bundle
    assert p.reproj_loss_type == LOSS_SOFT_L1 and p.reproj_loss_scale == 1.5 and p.reproj_loss_magnitude == 1.0
    assert p.depth_loss_type == LOSS_CAUCHY and 0 < p.n_dobs <= p.n_obs
    assert np.all(p.dobs_depth > 0) and np.all(p.dobs_magnitude > 0) and np.all(p.dobs_param > 0)
    assert set(ss) == set(order) and all(np.all(v == 0) for v in ss.values())
    # weights follow bundle_adjustment.py:153-161 with rob_std=2, multipliers 1
    im0 = sc.images[order[0]]
    sel = p.dobs_cam == 0
    var = p.dobs_depth[sel] ** 2 / p.dobs_magnitude[sel]
    np.testing.assert_allclose(p.dobs_param[sel], 2 * np.sqrt(var) / p.dobs_depth[sel], rtol=1e-12)
    assert sel.sum() <= len(im0.get_observation_point2D_idxs())


def test_scale_filter_and_param_multiplier(scene):
    sc, _, _ = scene
    opt = Optimizer({}, sc, None, backend=OracleBackend())
    b = _global_bundle(sc)
    f0, _ = opt._build_problem(b, False, True, mode="global", solve=False)
    f1, _ = opt._build_problem(b, False, True, mode="global", allow_scale_filter=True, param_multiplier=0.125, solve=False)
    assert f1.prob.n_dobs <= f0.prob.n_dobs
    opt.truncation_multiplier = 2.0
    f2, _ = opt._build_problem(b, False, True, mode="global", solve=False)
    np.testing.assert_allclose(f2.prob.dobs_param, 2.0 * f0.prob.dobs_param, rtol=1e-12)


def test_local_mode_pulls_outside_observations(scene):
    sc, _, _ = scene
    ids = sorted(sc.images.keys())
    ref = ids[3]
    pts = set(sc.images[ref].point3D_ids(sc.images[ref].get_observation_point2D_idxs()))
    bundle = {"ref_id": ref, "optim_ids": {ref, ids[2], ids[4]}, "pts3D": pts, "constpoints": set()}
    opt = Optimizer({}, sc, None, backend=OracleBackend())
    flat, _ = opt._build_problem(bundle, False, True, mode="local", solve=False)
    p = flat.prob
    n_cfg = 3
    assert p.n_cams > n_cfg and np.all(p.pose_const[n_cfg:] == 1)  # outside images, constant pose
    # explicit variable points are variable, the others observed by the bundle but with outside views are constant
    for pi, pid in enumerate(flat.point_ids):
        tl = sc.points3D[pid].track.length()
        nobs = int((p.obs_pt == pi).sum())
        assert p.pt_const[pi] == (1 if tl > nobs else 0)
        if pid in pts:
            assert p.pt_const[pi] == 0
    flat_g, _ = opt._build_problem(bundle, False, True, mode=None, solve=False)
    assert flat_g.prob.n_cams == n_cfg and flat_g.prob.pt_const.sum() > 0


def test_outside_observations_by_tracks_and_by_images_are_the_same_arrays(scene):
    """The local bundle's outside observations: walking every track (what pycolmap's C++ adjuster does) and sweeping the other
    images' observation lists with the bulk accessors must assemble the same flat problem, observation for observation."""
    sc, _, _ = scene
    ids = sorted(sc.images.keys())
    ref = ids[3]
    pts = set(sc.images[ref].point3D_ids(sc.images[ref].get_observation_point2D_idxs()))
    bundle = {"ref_id": ref, "optim_ids": {ref, ids[2], ids[4]}, "pts3D": pts, "constpoints": set()}
    flats = []
    for method in ("tracks", "images"):
        opt = Optimizer({}, sc, None, backend=OracleBackend())
        opt.outside_method = method
        flat, _ = opt._build_problem(bundle, False, True, mode="local", solve=False)
        flats.append(flat)
    a, b = flats
    assert a.prob.n_cams > 3 and a.prob.n_obs > 0
    assert list(a.image_ids) == list(b.image_ids)
    np.testing.assert_array_equal(a.point_ids, b.point_ids)
    for name in ("obs_cam", "obs_pt", "obs_xy", "pose_const", "pt_const", "cam_quat", "cam_t", "pts", "dobs_cam", "dobs_pt", "dobs_depth"):
        np.testing.assert_array_equal(getattr(a.prob, name), getattr(b.prob, name), err_msg=name)


def test_ba_improves_poses_and_writes_back_in_place(scene):
    prob, truth = make_scene(8, 400, True, seed=21)
    sc = scene_from_problem(prob, truth, seed=3)
    opt = Optimizer({}, sc, None, backend=OracleBackend())
    ids = list(_global_bundle(sc)["optim_ids"])
    quat_views = {i: sc.images[i].cam_from_world.rotation.quat for i in ids}
    xyz_view = sc.points3D[next(iter(sc.points3D))].xyz
    before_t = np.array([sc.images[i].cam_from_world.translation.copy() for i in ids])
    xyz_before = xyz_view.copy()
    result, ok = opt.ba(_global_bundle(sc), mode="global")
    assert ok is True and result.summary["final_cost"] < 0.5 * result.summary["initial_cost"]
    after_t = np.array([sc.images[i].cam_from_world.translation for i in ids])
    assert np.all(after_t[0] == before_t[0])            # first image is the gauge: untouched
    assert after_t[1][0] == before_t[1][0]              # second image: x translation fixed
    assert np.abs(after_t[2:] - before_t[2:]).max() > 1e-4
    assert quat_views[ids[2]] is sc.images[ids[2]].cam_from_world.rotation.quat  # same buffer, mutated in place
    assert np.any(xyz_view != xyz_before)


def test_refine_3d_points_keeps_poses(scene):
    prob, truth = make_scene(6, 250, True, seed=22)
    sc = scene_from_problem(prob, truth, seed=4)
    opt = Optimizer({}, sc, None, backend=OracleBackend())
    q0 = {i: im.cam_from_world.rotation.quat.copy() for i, im in sc.images.items()}
    res, ok = opt.refine_3d_points(_global_bundle(sc), depth_type="prior")
    assert ok and res.prob.depth_loss_type == LOSS_TRIVIAL and res.prob.pose_const.all()
    assert res.summary["reduced_dim"] == 0 and res.summary["final_cost"] < res.summary["initial_cost"]
    for i, im in sc.images.items():
        np.testing.assert_array_equal(im.cam_from_world.rotation.quat, q0[i])


def test_point_covs_and_zvars(scene):
    sc, _, _ = scene
    opt = Optimizer({}, sc, None, backend=OracleBackend())
    b = _global_bundle(sc)
    assert opt.calculate_point_covs(b) is None
    assert set(sc.point_covs.data) == b["pts3D"]
    c = next(iter(sc.point_covs.data.values()))
    assert c.shape == (3, 3) and np.allclose(c, c.T) and np.all(np.linalg.eigvalsh(c) > 0)
    im = next(iter(sc.images.values()))
    ids, zv = sc.point_covs.points_zvars(im)
    assert len(ids) == len(zv) and np.all(zv > 0)


def test_truncation_multiplier_and_shiftscale(scene):
    prob, truth = make_scene(6, 300, True, seed=23)
    sc = scene_from_problem(prob, truth, seed=5)
    opt = Optimizer({"min_truncation_mult": 0.5}, sc, None, backend=OracleBackend())
    opt.update_truncation_multiplier(list(sc.images.keys()))
    assert 0.5 <= opt.truncation_multiplier < 50
    for im in sc.images.values():  # a prior that is 2x too small must be rescaled by ~2
        im.depth.data_prior *= 0.5
    out, ok = opt.optimize_prior_shiftscale(_global_bundle(sc))
    assert ok and set(out) == set(sc.images)
    scales = np.array([s for _, s in out.values()])
    assert np.all(np.abs(scales / 2.0 - 1.0) < 0.1) and all(sh == 0.0 for sh, _ in out.values())


def test_unknown_conf_key_is_rejected(scene):
    sc, _, _ = scene
    with pytest.raises(KeyError):
        Optimizer({"not_a_key": 1}, sc, None, backend=OracleBackend())


def test_local_window_helper_equals_the_shims_local_assembly():
    """mpsfm_amd.synthetic.local_window (used by the full-size C5 GPU test) cuts the same reprojection problem out of a
    flat global problem as Optimizer._build_problem(mode="local") assembles from the scene objects."""
    from mpsfm_amd.synthetic import local_window

    prob, truth = make_scene(12, 900, True, seed=29)
    sc = scene_from_problem(prob, truth, seed=2)
    ids = sorted(sc.images)
    ref = ids[5]
    optim = {ids[4], ids[5], ids[6], ids[3]}
    bundle = {"ref_id": ref, "optim_ids": optim, "constpoints": set(),
              "pts3D": set(sc.images[ref].point3D_ids(sc.images[ref].get_observation_point2D_idxs()))}
    flat, _ = Optimizer({}, sc, None, backend=OracleBackend())._build_problem(bundle, False, True, mode="local", solve=False)
    window = [i - ids[0] for i in list(optim)]  # same order as the shim's list(bundle["optim_ids"])
    loc, cams, pts = local_window(prob, window, ref - ids[0])
    p = flat.prob
    assert [i - ids[0] for i in flat.image_ids[:4]] == window and sorted(i - ids[0] for i in flat.image_ids) == sorted(cams.tolist())
    assert p.n_obs == loc.n_obs and p.n_pts == loc.n_pts and p.gauge_axis_cam == loc.gauge_axis_cam == 1
    np.testing.assert_array_equal(p.pose_const[:4], loc.pose_const[:4])
    assert p.pose_const[4:].all() and loc.pose_const[4:].all()
    pid_of = sc._pid_of_problem_point
    const_shim = {pid: int(c) for pid, c in zip(flat.point_ids, p.pt_const)}
    const_loc = {pid_of[int(g)]: int(c) for g, c in zip(pts, loc.pt_const)}
    assert const_shim == const_loc and 0 < sum(const_loc.values()) < len(const_loc)

    def blocks(pr, cam_ids, pt_ids):
        return sorted((int(cam_ids[c]), int(pt_ids[q]), float(x), float(y)) for c, q, (x, y) in zip(pr.obs_cam, pr.obs_pt, pr.obs_xy))

    shim_blocks = blocks(p, [i - ids[0] for i in flat.image_ids], flat.point_ids)
    loc_blocks = blocks(loc, cams, [pid_of[int(g)] for g in pts])
    assert shim_blocks == loc_blocks


def _run_parts(nparts, reps):
    import ctypes as C

    from mpsfm_amd import capi

    L = capi.lib()
    L.mpsfm_debug_run_parts.restype = C.c_int64
    L.mpsfm_debug_run_parts.argtypes = [C.c_int32, C.c_int32]
    return int(L.mpsfm_debug_run_parts(nparts, reps))


def test_table_build_worker_pool_runs_every_part_once():
    """The persistent workers of the table build (csrc/ba_solver.hip: HostPool): every part of every job exactly once, for
    more parts than workers, for concurrent callers (the second finds the pool busy and starts plain threads) and in a forked
    child (which must not wait for the parent's workers)."""
    import os
    import threading

    for nparts in (1, 2, 7, 16, 61):
        assert _run_parts(nparts, 50) == 0
    out = []
    ths = [threading.Thread(target=lambda: out.append(_run_parts(9, 300))) for _ in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert out == [0, 0, 0, 0]
    pid = os.fork()
    if pid == 0:
        rc = 1
        try:
            rc = 0 if _run_parts(8, 20) == 0 else 2
        finally:
            os._exit(rc)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    assert _run_parts(8, 20) == 0


def test_table_build_host_blocks_are_recycled():
    """Large host blocks of the table build come back from the process-wide cache (csrc/ba_solver.hip: HostBlockCache);
    small ones bypass it."""
    import ctypes as C

    from mpsfm_amd import capi

    L = capi.lib()
    L.mpsfm_debug_host_cache.restype = C.c_int64
    L.mpsfm_debug_host_cache.argtypes = [C.c_int64, C.c_int32]
    assert L.mpsfm_debug_host_cache(3 << 20, 4) == 4      # 3 MB blocks (4 MB buckets): all four recycled
    assert L.mpsfm_debug_host_cache(3 << 20, 4) == 4      # and again from the same cache
    assert L.mpsfm_debug_host_cache(5 << 20, 2) == 2      # another bucket
    L.mpsfm_debug_host_cache(1000, 8)                      # below the cache's minimum: plain new / delete, any answer is fine
