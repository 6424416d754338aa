"""Full-size parity against the oracle at the BASELINE configurations:
  C3  the whole solve: iteration count, termination, cost trace, poses and points;
  C4  the reduced camera system (S, rhs), the dense solution and the model cost change of one LM step;
  C5  the whole solve, then the workload AS SPECIFIED: sliding local windows of 6 images (local mode: constant outside
      cameras and points, track < 15 rule — reference bundle_adjustment.py:88-91, mapper/base.py:442-474) each
      solved on the GPU and by the oracle from the same state, followed by the retriangulation numerics of the
      window's tracks.
The oracle needs 2 s (C3), 13 s (C5 solve) and 14 s (C4 reduced system) on the box's host cores."""

import numpy as np
import pytest

from mpsfm_amd import capi
from mpsfm_amd.problem import Tracks
from mpsfm_amd.synthetic import local_window, make_config
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def _assert_same_solve(sg, so, pg, po, state_atol):
    assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"]
    assert sg["num_successful_steps"] == so["num_successful_steps"]
    assert sg["trace_accepted"] == so["trace_accepted"]
    np.testing.assert_allclose(sg["trace_cost"], so["trace_cost"], rtol=1e-9)
    np.testing.assert_allclose(sg["trace_radius"], so["trace_radius"], rtol=1e-6)
    assert sg["initial_cost"] == pytest.approx(so["initial_cost"], rel=1e-12)
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-9)  # north-star bar: 1e-4
    np.testing.assert_allclose(pg.cam_t, po.cam_t, rtol=0, atol=state_atol)
    assert np.abs(np.abs(np.sum(pg.cam_quat * po.cam_quat, axis=1)) - 1.0).max() < 1e-10
    np.testing.assert_allclose(pg.pts, po.pts, rtol=0, atol=state_atol)


@pytest.mark.timeout(600)
def test_c3_full_solve_parity():
    prob, _ = make_config("C3")
    pg, po = prob.copy(), prob.copy()
    with capi.BAHandle(pg) as h:
        sg = h.solve()
        h.get_state()
    so = O.solve(po)
    _assert_same_solve(sg, so, pg, po, state_atol=1e-7)
    assert sg["num_residual_blocks"] == prob.n_obs + prob.n_dobs == so["num_residual_blocks"]
    assert sg["num_residual_evals"] == so["num_residual_evals"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["C4", "C5"])
def test_reduced_system_and_step_parity_large(name):
    prob, _ = make_config(name)
    radius = 1e4
    ref = O.reduced_system(prob, radius=radius)
    with capi.BAHandle(prob.copy()) as h:
        assert h.reduced_dim == ref["S"].shape[0] == 6 * (prob.n_cams - 1)
        h.sweep_once(radius)
        S, rhs = h.reduced_system()
        scale = np.abs(ref["S"]).max()
        # block-wise: every 6x6 block against the largest entry of its block row (a global bound would hide small blocks)
        n = S.shape[0]
        nb = n // 6
        diff = np.abs(S - ref["S"])[: nb * 6, : nb * 6].reshape(nb, 6, nb, 6).max(axis=(1, 3))
        rowmax = np.abs(ref["S"])[: nb * 6, : nb * 6].reshape(nb, 6, nb, 6).max(axis=(1, 2, 3))
        assert (diff <= 1e-11 * rowmax[:, None] + 1e-14 * scale).all(), float((diff / rowmax[:, None]).max())
        np.testing.assert_allclose(rhs, ref["rhs"], rtol=0, atol=1e-11 * np.abs(ref["rhs"]).max())
        # sparsity the static pair tables imply: blocks the oracle leaves exactly zero are exactly zero here too
        zero_ref = np.abs(ref["S"])[: nb * 6, : nb * 6].reshape(nb, 6, nb, 6).max(axis=(1, 3)) == 0
        assert (np.abs(S)[: nb * 6, : nb * 6].reshape(nb, 6, nb, 6).max(axis=(1, 3))[zero_ref] == 0).all()
        h.dense_solve_once()
        y = h.dense_solution()
        np.testing.assert_allclose(y, ref["yc"], rtol=0, atol=1e-8 * np.abs(ref["yc"]).max())
        print(f"{name}: n = {n}, zero blocks {zero_ref.mean():.3f}, max |y - y_ref| / max|y| = {np.abs(y - ref['yc']).max() / np.abs(ref['yc']).max():.2e}")


@pytest.mark.timeout(900)
def test_c5_full_solve_then_local_windows_as_specified():
    prob, _ = make_config("C5")
    pg, po = prob.copy(), prob.copy()
    with capi.BAHandle(pg) as h:
        sg = h.solve()
        h.get_state()
    so = O.solve(po)
    _assert_same_solve(sg, so, pg, po, state_atol=1e-7)
    # ---- incremental part: local bundle adjustments around newly "registered" images, sliding over the orbit ----------
    rng = np.random.default_rng(0)
    glob_g, glob_o = pg, po  # both sides continue from their own global optimum (equal to 1e-7)
    for step, ref_cam in enumerate((20, 21, 22, 150, 299)):
        window = [ref_cam] + [(ref_cam + d) % prob.n_cams for d in (-1, 1, -2, 2, -3)]  # local_ba_num_images = 6
        # the new image's pose and its points are what registration + triangulation just produced: perturb them
        for gl in (glob_g, glob_o):
            r = np.random.default_rng(100 + step)
            gl.cam_t[ref_cam] += r.normal(0, 0.02, 3)
            seen = np.unique(gl.obs_pt[gl.obs_cam == ref_cam])
            gl.pts[seen] += r.normal(0, 0.02, (len(seen), 3))
        lg, cams, pts = local_window(glob_g, window, ref_cam)
        lo, cams_o, pts_o = local_window(glob_o, window, ref_cam)
        np.testing.assert_array_equal(cams, cams_o)
        np.testing.assert_array_equal(pts, pts_o)
        assert lg.n_cams > 6 and lg.pose_const[:6].tolist() == [1, 0, 0, 0, 0, 0] and lg.pose_const[6:].all()
        assert 0 < lg.pt_const.sum() < lg.n_pts and lg.gauge_axis_cam == 1
        s_g, s_o = capi.ba_solve(lg), O.solve(lo)
        _assert_same_solve(s_g, s_o, lg, lo, state_atol=1e-6)
        assert s_g["reduced_dim"] == 30 and s_g["final_cost"] < s_g["initial_cost"]
        np.testing.assert_array_equal(lg.cam_quat[0], glob_g.cam_quat[window[0]])  # constant pose untouched
        np.testing.assert_array_equal(lg.cam_quat[6:], glob_g.cam_quat[cams[6:]])
        for gl, lc in ((glob_g, lg), (glob_o, lo)):  # in-place write-back, as Ceres does
            gl.cam_quat[cams], gl.cam_t[cams] = lc.cam_quat, lc.cam_t
            var = lc.pt_const == 0
            gl.pts[pts[var]] = lc.pts[var]
        # retriangulation numerics of the window's tracks at the refined poses: HIP vs tri oracle
        order = np.argsort(glob_g.obs_pt, kind="stable") if step == 0 else order
        start = np.searchsorted(glob_g.obs_pt[order], np.arange(glob_g.n_pts + 1)) if step == 0 else start
        pick = pts[rng.choice(len(pts), min(4000, len(pts)), replace=False)]
        pick.sort()
        cnt = (start[pick + 1] - start[pick])
        idx = np.concatenate([np.arange(start[p], start[p + 1]) for p in pick])
        sub = Tracks(glob_g.cam_quat, glob_g.cam_t, glob_g.cam_intr, glob_g.cam_intr_idx, np.concatenate([[0], np.cumsum(cnt)]),
                     glob_g.obs_cam[order][idx], glob_g.obs_xy[order][idx])
        x_g, x_o = capi.triangulate_tracks(sub), O.triangulate_tracks(sub)
        # same inputs to both; a low-parallax track amplifies the rounding of the 4x4 eigen-solve (observed 1.0e-8 absolute = 6e-10
        # relative on one coordinate of 12000)
        np.testing.assert_allclose(x_g, x_o, rtol=1e-8, atol=1e-8)
        a_g, e_g, f_g = capi.filter_tracks(sub, x_g)
        a_o, e_o, f_o = O.filter_tracks(sub, x_g)
        np.testing.assert_allclose(a_g, a_o, rtol=0, atol=1e-11)
        np.testing.assert_allclose(e_g, e_o, rtol=1e-9, atol=1e-11)
        np.testing.assert_array_equal(f_g, f_o)
        # the decisions the mapper takes from these numbers (retri_min_angle 1.5 deg, max reprojection error 4 px): exact
        np.testing.assert_array_equal(a_g < np.deg2rad(1.5), a_o < np.deg2rad(1.5))
        np.testing.assert_array_equal(e_g > 16.0, e_o > 16.0)
        print(f"window around camera {ref_cam}: {lg.n_cams} cameras ({lg.n_cams - 6} constant outside), {lg.n_pts} landmarks "
              f"({int(lg.pt_const.sum())} constant), {lg.n_obs + lg.n_dobs} blocks, {s_g['num_iterations']} LM iterations, "
              f"{1e3 * s_g['time_total_s']:.2f} ms")
