"""GPU parity tests beyond the core solve: the Optimizer shim end to end, point covariances,
triangulation numerics, landmark sharding on device buffers, edge cases and full-size properties."""

import threading

import numpy as np
import pytest

from mpsfm_amd import capi
from mpsfm_amd.dist import shard_problem
from mpsfm_amd.problem import BAProblem, Tracks
from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from mpsfm_amd.sfm.mapper.triangulator import MpsfmTriangulator, track_quality, triangulate_points
from numpy_scene import scene_from_problem
from mpsfm_amd.synthetic import R_from_quat, make_config, make_scene
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


from backends import OracleBackend  # noqa: E402


def _bundle(sc):
    return {"optim_ids": set(sc.images.keys()), "pts3D": set(sc.points3D.keys()), "constpoints": set()}


@pytest.mark.parametrize("mode", ["global", "local"])
def test_optimizer_shim_matches_oracle_backend(mode):
    prob, truth = make_scene(8, 400, True, seed=21)
    sc_g, sc_o = scene_from_problem(prob, truth, seed=3), scene_from_problem(prob, truth, seed=3)
    og, oo = Optimizer({}, sc_g, None), Optimizer({}, sc_o, None, backend=OracleBackend())
    if mode == "global":
        bg, bo = _bundle(sc_g), _bundle(sc_o)
    else:
        ids = sorted(sc_g.images)
        pts = set(sc_g.images[ids[3]].point3D_ids(sc_g.images[ids[3]].get_observation_point2D_idxs()))
        bg = {"ref_id": ids[3], "optim_ids": {ids[2], ids[3], ids[4]}, "pts3D": pts, "constpoints": set()}
        bo = dict(bg)
    for o, b in ((og, bg), (oo, bo)):
        o.calculate_point_covs(b)
        o.update_truncation_multiplier(list(b["optim_ids"]))
    assert og.truncation_multiplier == pytest.approx(oo.truncation_multiplier, rel=1e-12)
    rg, _ = og.ba(bg, mode=mode, allow_scale_filter=True)
    ro, _ = oo.ba(bo, mode=mode, allow_scale_filter=True)
    assert rg.summary["final_cost"] == pytest.approx(ro.summary["final_cost"], rel=1e-8)
    for i in sc_g.images:
        np.testing.assert_allclose(sc_g.images[i].cam_from_world.translation, sc_o.images[i].cam_from_world.translation, atol=1e-6)
        np.testing.assert_allclose(np.abs(sc_g.images[i].cam_from_world.rotation.quat @ sc_o.images[i].cam_from_world.rotation.quat), 1, atol=1e-10)
    for p in sc_g.points3D:
        np.testing.assert_allclose(sc_g.points3D[p].xyz, sc_o.points3D[p].xyz, atol=1e-6)
    for p in bg["pts3D"]:
        np.testing.assert_allclose(sc_g.point_covs.data[p], sc_o.point_covs.data[p], rtol=1e-9, atol=1e-18)
    rg, _ = og.refine_3d_points(bg, depth_type="prior")
    ro, _ = oo.refine_3d_points(bo, depth_type="prior")
    assert rg.summary["final_cost"] == pytest.approx(ro.summary["final_cost"], rel=1e-9)


def test_point_covs_match_oracle():
    prob, _ = make_scene(12, 2000, False, seed=7)
    prob.reproj_loss_magnitude = 0.25
    c_g, c_o = capi.point_covs(prob), O.point_covs(prob)
    np.testing.assert_allclose(c_g, c_o, rtol=1e-9, atol=1e-18)
    assert np.all(np.linalg.eigvalsh(c_g) > 0)


def _tracks_of(prob, truth):
    order = np.argsort(prob.obs_pt, kind="stable")
    start = np.searchsorted(prob.obs_pt[order], np.arange(prob.n_pts + 1))
    return Tracks(truth["cam_quat"], truth["cam_t"], prob.cam_intr, prob.cam_intr_idx, start, prob.obs_cam[order], prob.obs_xy[order])


def test_triangulation_numerics_match_oracle_and_truth():
    prob, truth = make_scene(15, 3000, False, seed=9, outlier_frac=0.0)
    tr = _tracks_of(prob, truth)
    x_g, x_o = capi.triangulate_tracks(tr), O.triangulate_tracks(tr)
    np.testing.assert_allclose(x_g, x_o, rtol=0, atol=1e-9)
    assert np.median(np.linalg.norm(x_g - truth["pts"], axis=1)) < 0.05  # 1 px noise, fp16-rounded pixels
    a_g, e_g, f_g = capi.filter_tracks(tr, x_g)
    a_o, e_o, f_o = O.filter_tracks(tr, x_g)
    np.testing.assert_allclose(a_g, a_o, rtol=0, atol=1e-12)
    np.testing.assert_allclose(e_g, e_o, rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(f_g, f_o)
    assert f_g.all() and 0 < a_g.max() <= np.pi / 2


def test_triangulator_lifts_low_parallax_points():
    prob, truth = make_scene(6, 200, True, seed=11)
    sc = scene_from_problem(prob, truth, seed=1)
    tri = MpsfmTriangulator({"colmap_options": {}}, sc, None)
    ids = list(sc.points3D)
    ang, err, front = track_quality(sc, ids)
    assert ang.shape == (len(ids),) and front.all()
    thr = np.rad2deg(np.median(ang))
    n_before = len(sc.points3D)
    new_ids = tri.lift_low_parallax(ids, thr)
    assert 0 < len(new_ids) <= (ang < np.deg2rad(thr)).sum() and len(sc.points3D) <= n_before
    for pid in new_ids:  # a lifted point sits on the viewing ray of its first activated image at the sampled depth
        el = sc.points3D[pid].track.elements[0]
        im = sc.images[el.image_id]
        z = (im.cam_from_world * sc.points3D[pid].xyz[None])[0, 2]
        assert z > 0
    xyz = triangulate_points(sc, ids[:5] if all(i in sc.points3D for i in ids[:5]) else list(sc.points3D)[:5])
    assert xyz.shape == (5, 3)
    with pytest.raises(ValueError):  # neither a correspondence graph nor an engine: nothing to walk
        tri.complete_and_merge_all_tracks()


def test_landmark_sharded_solve_on_device_buffers():
    """Two shards on one GPU, each with its own handle and thread; the hook sums the device
    buffers with torch (stands in for the RCCL all-reduce across ranks)."""
    import ctypes as C

    import torch

    from mpsfm_amd.dist import _DevView
    from mpsfm_amd.problem import ALLREDUCE_FN

    prob, _ = make_scene(12, 1500, True, seed=13)
    ref = prob.copy()
    s_ref = capi.ba_solve(ref)
    world = 2
    bar = threading.Barrier(world)
    slots, results = [None] * world, [None] * world
    lock = threading.Lock()

    def make_hook(rank):
        def cb(user, buf, count, on_device, stream):
            ptr = C.addressof(buf.contents)
            if on_device:
                torch.cuda.synchronize()
                t = torch.as_tensor(_DevView(ptr, int(count)), device="cuda")
            else:
                t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(count),)))
            slots[rank] = t
            bar.wait()
            total = slots[0].clone().to(t.device) + slots[1].clone().to(t.device)
            bar.wait()
            t.copy_(total)
            if on_device:
                torch.cuda.synchronize()
            bar.wait()
            return 0
        return ALLREDUCE_FN(cb)

    shards = [shard_problem(prob, r, world) for r in range(world)]

    def run(rank):
        fn = make_hook(rank)
        opts = capi.default_options()
        opts.allreduce = fn
        results[rank] = capi.ba_solve(shards[rank][0], opts)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=240) for t in th]
    assert all(r is not None for r in results)
    for r in range(world):
        assert results[r]["num_iterations"] == s_ref["num_iterations"]
        assert results[r]["final_cost"] == pytest.approx(s_ref["final_cost"], rel=1e-9)
        lo, hi = shards[r][1]
        np.testing.assert_allclose(shards[r][0].pts, ref.pts[lo:hi], atol=1e-7)
        np.testing.assert_allclose(shards[r][0].cam_t, ref.cam_t, atol=1e-7)


def test_edge_cases():
    base, _ = make_scene(4, 60, True, seed=17)
    # no residual blocks at all
    empty = BAProblem(base.cam_quat, base.cam_t, base.pts, base.cam_intr, base.cam_intr_idx, base.pose_const, base.pt_const,
                      np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    s = capi.ba_solve(empty)
    assert s["termination"] == "no_variables" and s["final_cost"] == 0.0
    # everything constant: cost only, equals the oracle's
    allc = base.copy()
    allc.pose_const[:] = 1
    allc.pt_const[:] = 1
    sg, so = capi.ba_solve(allc.copy()), O.solve(allc.copy())
    assert sg["termination"] == "no_variables" and sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-12)
    # one free camera, constant points (pose refinement); unreferenced points stay untouched
    pose = base.copy()
    pose.pt_const[:] = 1
    pose.pose_const[:] = [1, 1, 1, 0]
    pose.gauge_axis_cam = -1
    pg, po = pose.copy(), pose.copy()
    sg, so = capi.ba_solve(pg), O.solve(po)
    assert sg["reduced_dim"] == 6 and sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-9)
    np.testing.assert_array_equal(pg.pts, pose.pts)
    # a point behind the camera in a log-depth block: the initial point cannot be evaluated
    bad = base.copy()
    bad.pts[int(bad.dobs_pt[0])] = -50 * bad.pts[int(bad.dobs_pt[0])] - 100.0
    with pytest.raises(capi.MpsfmHipError) as e:
        capi.ba_solve(bad)
    assert e.value.code == -5


def test_long_tracks_are_swept_by_their_own_workgroups():
    """Tracks longer than a chunk (> 256 records or > 254 cameras): every camera sees landmarks 0..2, some
    cameras twice (two reprojection blocks of one camera on one landmark) and with depth priors."""
    rng = np.random.default_rng(5)
    prob, truth = make_scene(260, 400, True, seed=19)
    R = R_from_quat(truth["cam_quat"])
    oc, op, oxy, dd, dm, da = [], [], [], [], [], []
    for pt in range(3):
        Xc = R @ truth["pts"][pt] + truth["cam_t"]
        uv = np.stack([1200 * Xc[:, 0] / Xc[:, 2] + 800, 1200 * Xc[:, 1] / Xc[:, 2] + 600], 1) + rng.normal(0, 1, (260, 2))
        oc.append(np.arange(260)); op.append(np.full(260, pt)); oxy.append(uv)
        d = Xc[:, 2] * np.exp(rng.normal(0, 0.0263, 260))
        var = np.maximum((0.0263 * d) ** 2, 0.02**2)
        dd.append(d); dm.append(d**2 / var); da.append(2 * np.sqrt(var) / d)
    long_prob = BAProblem(
        prob.cam_quat, prob.cam_t, prob.pts, prob.cam_intr, prob.cam_intr_idx, prob.pose_const, prob.pt_const,
        np.concatenate([prob.obs_cam] + oc).astype(np.int32), np.concatenate([prob.obs_pt] + op).astype(np.int32),
        np.concatenate([prob.obs_xy] + oxy), gauge_axis_cam=prob.gauge_axis_cam,
        dobs_cam=np.concatenate([prob.dobs_cam] + oc).astype(np.int32), dobs_pt=np.concatenate([prob.dobs_pt] + op).astype(np.int32),
        dobs_depth=np.concatenate([prob.dobs_depth] + dd), dobs_magnitude=np.concatenate([prob.dobs_magnitude] + dm),
        dobs_param=np.concatenate([prob.dobs_param] + da), depth_loss_type=prob.depth_loss_type)
    ref = O.reduced_system(long_prob, radius=50.0)
    with capi.BAHandle(long_prob.copy()) as h:
        cr, cd = h.eval_cost()
        ocr, ocd = O.eval_cost(long_prob)
        assert cr == pytest.approx(ocr, rel=1e-12) and cd == pytest.approx(ocd, rel=1e-12)
        h.sweep_once(50.0)
        S, rhs = h.reduced_system()
        np.testing.assert_allclose(S, ref["S"], rtol=0, atol=1e-10 * np.abs(ref["S"]).max())
        np.testing.assert_allclose(rhs, ref["rhs"], rtol=0, atol=1e-10 * np.abs(ref["rhs"]).max())
    pg, po = long_prob.copy(), long_prob.copy()
    sg, so = capi.ba_solve(pg), O.solve(po)
    assert sg["num_iterations"] == so["num_iterations"]
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8)
    np.testing.assert_allclose(pg.pts[:3], po.pts[:3], atol=1e-6)
    np.testing.assert_allclose(pg.cam_t, po.cam_t, atol=1e-6)
    # a constant long-track landmark only feeds the camera blocks
    cp = long_prob.copy()
    cp.pt_const[:3] = 1
    pg, po = cp.copy(), cp.copy()
    sg, so = capi.ba_solve(pg), O.solve(po)
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8)
    np.testing.assert_array_equal(pg.pts[:3], cp.pts[:3])


def test_full_size_properties_c3():
    """BASELINE config C3 (200 cameras / 150k landmarks): size-independent properties."""
    prob, _ = make_config("C3")
    with capi.BAHandle(prob) as h:
        c0 = sum(h.eval_cost())
        s1 = h.solve()
        c1 = sum(h.eval_cost())
        assert s1["initial_cost"] == pytest.approx(c0, rel=1e-10)
        assert s1["final_cost"] == pytest.approx(c1, rel=1e-10)          # summary cost is the cost of the returned state
        tc = np.array(s1["trace_cost"])
        assert np.all(np.diff(tc) <= 1e-9 * tc[:-1]) and s1["final_cost"] < 0.15 * s1["initial_cost"]
        assert s1["termination"] == "function_tolerance" and s1["num_iterations"] <= 50
        assert s1["num_residual_blocks"] == prob.n_obs + prob.n_dobs
        h.reset_state()
        s2 = h.solve()                                                    # reset + solve again: same trajectory
        assert s2["num_iterations"] == s1["num_iterations"]
        assert s2["final_cost"] == pytest.approx(s1["final_cost"], rel=1e-10)
        h.get_state()
        # gauge: camera 0 untouched, camera 1 keeps its x translation
        ref, _ = make_config("C3")
        np.testing.assert_array_equal(prob.cam_quat[0], ref.cam_quat[0])
        assert prob.cam_t[1, 0] == ref.cam_t[1, 0]
        # re-solving from the optimum stops at once
        h.set_state(prob)
        s3 = h.solve()
        assert s3["num_iterations"] <= 2 and s3["final_cost"] <= s1["final_cost"] * (1 + 1e-9)


@pytest.mark.parametrize("variant", ["trivial_reproj", "cauchy_reproj", "kp_std_2", "two_intrinsics", "shift_logscale", "no_jacobi"])
def test_solver_variants_match_oracle(variant):
    prob, _ = make_scene(9, 700, True, seed=41)
    opts_g, opts_o = capi.default_options(), O.default_options()
    if variant == "trivial_reproj":
        prob.reproj_loss_type = 0
    elif variant == "cauchy_reproj":
        prob.reproj_loss_type, prob.reproj_loss_scale = 2, 3.0
    elif variant == "kp_std_2":
        prob.reproj_loss_scale, prob.reproj_loss_magnitude = 3.0, 0.25
    elif variant == "two_intrinsics":
        prob.cam_intr = np.array([[1200.0, 1200.0, 800.0, 600.0], [1210.0, 1190.0, 805.0, 598.0]])
        prob.cam_intr_idx = (np.arange(prob.n_cams) % 2).astype(np.int32)
    elif variant == "shift_logscale":
        prob.shift_logscale = np.stack([np.full(prob.n_cams, 0.01), np.linspace(-0.02, 0.02, prob.n_cams)], 1)
    elif variant == "no_jacobi":
        opts_g.jacobi_scaling = 0
        opts_o.jacobi_scaling = 0
    pg, po = prob.copy(), prob.copy()
    sg, so = capi.ba_solve(pg, opts_g), O.solve(po, opts_o)
    assert sg["initial_cost"] == pytest.approx(so["initial_cost"], rel=1e-12)
    assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"]
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8)
    np.testing.assert_allclose(pg.pts, po.pts, atol=1e-6)
    np.testing.assert_allclose(pg.cam_t, po.cam_t, atol=1e-6)


def test_c2_reprojection_only_full_size():
    """BASELINE config C2: 50 cameras / 20k landmarks, reprojection only — full parity with the oracle."""
    prob, _ = make_config("C2")
    pg, po = prob.copy(), prob.copy()
    sg, so = capi.ba_solve(pg), O.solve(po)
    assert sg["num_iterations"] == so["num_iterations"]
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8)
    n = min(len(sg["trace_cost"]), len(so["trace_cost"]))
    np.testing.assert_allclose(sg["trace_cost"][:n], so["trace_cost"][:n], rtol=1e-9)
    # a few two-view landmarks with almost no parallax sit ~1e6 away: compare relatively
    np.testing.assert_allclose(pg.pts, po.pts, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("speculate", ["1", "0"])
def test_max_iterations_and_tight_tolerances(speculate, monkeypatch):
    """Termination by the iteration limit right after an accepted step (the state must be that step's candidate: the
    device-side loop copies it in the launch that FOLLOWS the decision), by tight tolerances, and by the radius limit; with
    the host one iteration ahead of the device (default) and waiting for every decision (MPSFM_LM_SPECULATE=0)."""
    monkeypatch.setenv("MPSFM_LM_SPECULATE", speculate)
    prob, _ = make_scene(7, 400, True, seed=43)
    for kw in (dict(max_num_iterations=3), dict(max_num_iterations=1), dict(function_tolerance=1e-12, parameter_tolerance=1e-12, max_num_iterations=80),
               dict(min_trust_region_radius=2e4), dict(max_num_iterations=0)):
        pg, po = prob.copy(), prob.copy()
        sg, so = capi.ba_solve(pg, capi.default_options(**kw)), O.solve(po, O.default_options(**kw))
        assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"], kw
        assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8), kw
        assert sg["trace_accepted"] == so["trace_accepted"], kw
        np.testing.assert_allclose(pg.cam_t, po.cam_t, rtol=0, atol=1e-9, err_msg=str(kw))
        np.testing.assert_allclose(pg.cam_quat, po.cam_quat, rtol=0, atol=1e-9, err_msg=str(kw))
        np.testing.assert_allclose(pg.pts, po.pts, rtol=0, atol=1e-8, err_msg=str(kw))


@pytest.mark.parametrize("collective", ["rccl", "hook", "rccl-fails"])
def test_rccl_single_rank(collective):
    """bench.py with a one-rank NCCL(RCCL) process group.  "rccl": the library's own communicator (ncclCommInitRank from a
    unique id, ncclAllReduce on the solver's stream, no Python in the LM loop); "hook": the torch.distributed hook with
    the zero-copy tensor view and the shared stream; "rccl-fails": the native path's first solve is made to fail, the run
    must continue on the hook by itself.  Same result as the plain single-GPU solve."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPSFM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MPSFM_BENCH_COLLECTIVE="hook" if collective == "hook" else "rccl")
    if collective == "rccl-fails":
        env["MPSFM_BENCH_INJECT_NATIVE_FAILURE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "C2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--kernel-reps", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    ref, _ = make_config("C2")
    s = capi.ba_solve(ref)
    assert d["solve"]["final_cost"] == pytest.approx(s["final_cost"], rel=1e-9)
    assert d["solve"]["lm_iterations"] == s["num_iterations"]
    assert ("native RCCL" if collective == "rccl" else "hook") in d["config"]["parallelism"]
    if collective == "rccl-fails":
        assert "falling back" in r.stderr


@pytest.mark.timeout(600)
@pytest.mark.parametrize("gtol", [None, 5.0e3, "fixed"])
def test_two_process_sharded_solve_on_one_gpu(tmp_path, gtol):
    """Two OS processes, one landmark shard each, both on cuda:0; the reduced system is summed across them on
    the DEVICE buffers through mpsfm_amd.dist.make_torch_allreduce (gloo here: RCCL does not allow two ranks
    on one GPU).  The library runs on a stream of its own, so this fails if the hook does not order the
    collective with that stream.  Must reproduce the single-process trajectory — also when the solve ends on the
    gradient tolerance (gtol case): the landmark-gradient maximum is a MAX over ranks inside a SUM all-reduce (per-rank
    slots), a sum of the per-rank maxima would stop later than the single-process solve.  "fixed": constant cameras and
    landmarks give fixed blocks, whose cost (and the landmark part of the state norm) is summed over the ranks on the device
    in front of the loop."""
    import json
    import os
    import socket
    import subprocess
    import sys

    seed, world = 77, 2
    fixed = gtol == "fixed"
    if fixed:
        gtol = None
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_gpu_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), OMP_NUM_THREADS="2",
                   MPSFM_TEST_GTOL="" if gtol is None else repr(gtol), MPSFM_TEST_FIXED="1" if fixed else "")
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), str(seed)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(world)]
    if not all(r["supported"] for r in res):
        pytest.skip("this torch build cannot all-reduce device tensors with gloo")
    ref, _ = make_scene(12, 6000, True, seed=seed)
    if fixed:
        ref.pose_const[2:5] = 1
        ref.pt_const[::4] = 1
    s = capi.ba_solve(ref, capi.default_options(**({} if gtol is None else {"gradient_tolerance": gtol})))
    if fixed:
        assert s["fixed_cost"] > 0.0
    assert s["termination"] == ("function_tolerance" if gtol is None else "gradient_tolerance")
    if gtol is not None:
        assert 2 <= s["num_iterations"] < 12
    for r, o in enumerate(res):
        assert o["termination"] == s["termination"]
        assert o["iters"] == s["num_iterations"] and o["nblocks"] == s["num_residual_blocks"]
        assert o["initial_cost"] == pytest.approx(s["initial_cost"], rel=1e-12)
        assert o["final_cost"] == pytest.approx(s["final_cost"], rel=1e-9)
        np.testing.assert_allclose(o["trace"], s["trace_cost"], rtol=1e-9)
        st = np.load(tmp_path / f"state{r}.npz")
        np.testing.assert_allclose(st["cam_t"], ref.cam_t, atol=1e-7)
        np.testing.assert_allclose(st["pts"], ref.pts[o["lo"]:o["hi"]], atol=1e-6)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["C4", "C5"])
def test_full_size_properties_large_configs(name):
    """BASELINE configs 4 (1000 cameras / 800k landmarks) and 5 (300 cameras, ~2 M observations) on one GPU:
    size-independent properties of the solve, then the retriangulation numerics over every track of the
    refined scene with a sample checked against the oracle."""
    prob, truth = make_config(name)
    ref_q0, ref_t1 = prob.cam_quat[0].copy(), prob.cam_t[1, 0]
    if name == "C5":
        assert prob.n_obs > 1_800_000
    with capi.BAHandle(prob) as h:
        c0 = sum(h.eval_cost())
        s = h.solve()
        c1 = sum(h.eval_cost())
        assert s["initial_cost"] == pytest.approx(c0, rel=1e-10) and s["final_cost"] == pytest.approx(c1, rel=1e-10)
        tc = np.array(s["trace_cost"])
        assert np.all(np.diff(tc) <= 1e-9 * tc[:-1]) and s["final_cost"] < 0.2 * s["initial_cost"]
        assert s["termination"] in ("function_tolerance", "parameter_tolerance") and s["num_iterations"] <= 50
        assert s["num_residual_blocks"] == prob.n_obs + prob.n_dobs and s["reduced_dim"] == 6 * (prob.n_cams - 1)
        h.get_state()
        np.testing.assert_array_equal(prob.cam_quat[0], ref_q0)
        assert prob.cam_t[1, 0] == ref_t1
        print(f"{name}: {s['num_iterations']} LM iterations, {1e3 * s['time_total_s']:.1f} ms, dense {1e3 * s['time_dense_s']:.1f} ms, "
              f"sweep {1e3 * s['time_linearize_s']:.1f} ms")
    # retriangulation numerics on all tracks with the refined poses
    order = np.argsort(prob.obs_pt, kind="stable")
    start = np.searchsorted(prob.obs_pt[order], np.arange(prob.n_pts + 1))
    tr = Tracks(prob.cam_quat, prob.cam_t, prob.cam_intr, prob.cam_intr_idx, start, prob.obs_cam[order], prob.obs_xy[order])
    x = capi.triangulate_tracks(tr)
    ang, err, front = capi.filter_tracks(tr, x)
    assert x.shape == (prob.n_pts, 3) and np.isfinite(x).mean() > 0.999 and front.mean() > 0.95
    assert np.nanmedian(np.linalg.norm(x - prob.pts, axis=1)) < 0.1
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(prob.n_pts, 3000, replace=False))
    cnt = np.diff(start)[pick]
    s2 = np.concatenate([[0], np.cumsum(cnt)])
    idx = np.concatenate([np.arange(start[p], start[p + 1]) for p in pick])
    sub = Tracks(prob.cam_quat, prob.cam_t, prob.cam_intr, prob.cam_intr_idx, s2, tr.el_cam[idx], tr.el_xy[idx])
    x_o = O.triangulate_tracks(sub)
    np.testing.assert_allclose(x[pick], x_o, rtol=0, atol=1e-8)
    a_o, e_o, f_o = O.filter_tracks(sub, x[pick])
    np.testing.assert_allclose(ang[pick], a_o, rtol=0, atol=1e-11)
    np.testing.assert_allclose(err[idx], e_o, rtol=1e-9, atol=1e-11)  # reprojection errors are per track element
    np.testing.assert_array_equal(front[idx], f_o)                      # so are the cheirality flags


@pytest.mark.parametrize("nb", [1, 2, 3, 5])
def test_outer_panel_cholesky_matches_oracle(nb, monkeypatch):
    """The dense factorisation in outer panels of `nb` tile columns (used by default above 64 tile columns, forced here
    through MPSFM_CHOL_NB on a 9-tile system): same LM trajectory as the oracle, same dense solution as the plain path."""
    prob, _ = make_scene(48, 5000, True, seed=61)
    with capi.BAHandle(prob.copy()) as h:
        h.sweep_once(1e4)
        h.dense_solve_once()
        y_plain = h.dense_solution()
    monkeypatch.setenv("MPSFM_CHOL_NB", str(nb))
    with capi.BAHandle(prob.copy()) as h:
        assert (h.reduced_dim + 31) // 32 == 9
        h.sweep_once(1e4)
        h.dense_solve_once()
        y_nb = h.dense_solution()
    np.testing.assert_allclose(y_nb, y_plain, rtol=1e-9, atol=1e-12 * np.abs(y_plain).max())
    pg, po = prob.copy(), prob.copy()
    sg, so = capi.ba_solve(pg), O.solve(po)
    assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"]
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8)


def test_tracks_of_up_to_254_cameras_stay_on_the_chunked_path():
    """Landmarks seen by 65..254 cameras (dense matching) are swept in chunks like short tracks (local camera list of 254
    entries); only beyond that, or beyond 256 records, the per-landmark kernels take over.  Parity of the reduced
    system and of the solve on tracks of ~100 cameras."""
    prob, _ = make_scene(200, 400, True, seed=23, max_track=150, track_mean=100.0)
    tl = np.bincount(prob.obs_pt)
    assert tl.max() > 120 and (tl > 64).mean() > 0.9
    ref = O.reduced_system(prob, radius=1e3)
    with capi.BAHandle(prob.copy()) as h:
        h.sweep_once(1e3)
        S, rhs = h.reduced_system()
        np.testing.assert_allclose(S, ref["S"], rtol=0, atol=1e-10 * np.abs(ref["S"]).max())
        np.testing.assert_allclose(rhs, ref["rhs"], rtol=0, atol=1e-10 * np.abs(ref["rhs"]).max())
    pg, po = prob.copy(), prob.copy()
    sg, so = capi.ba_solve(pg), O.solve(po)
    assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"]
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-8)
    np.testing.assert_allclose(pg.cam_t, po.cam_t, atol=1e-6)


def test_inverse_propagation_equals_grouped_back_substitution(monkeypatch):
    """y = L^-T z through the accumulators of L^-1 built during the factorisation (default up to 64 tile columns) against
    the grouped back substitution (MPSFM_CHOL_INVERSE=0) and numpy on the same reduced system, at two dampings."""
    prob, _ = make_scene(60, 6000, True, seed=71)
    ys = {}
    for inverse in ("1", "0"):
        monkeypatch.setenv("MPSFM_CHOL_INVERSE", inverse)
        with capi.BAHandle(prob.copy()) as h:
            for radius in (1e4, 1e-1):
                h.sweep_once(radius)
                h.dense_solve_once()
                y = h.dense_solution()
                S, rhs = h.reduced_system()
                if inverse == "1":
                    y_np = np.linalg.solve(S, rhs)
                    np.testing.assert_allclose(y, y_np, rtol=0, atol=1e-9 * np.abs(y_np).max())
                ys[(inverse, radius)] = y
    for radius in (1e4, 1e-1):
        a, b = ys[("1", radius)], ys[("0", radius)]
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-10 * np.abs(b).max())


def test_aborted_solve_does_not_hand_live_blocks_to_another_handle():
    """free_handle waits for the handle's streams before its blocks go back to the process-wide
    cache.  A solve aborted by a failing all-reduce hook on one thread, a clean solver on another, MPSFM_POISON=1."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPSFM_POISON="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_poison_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
