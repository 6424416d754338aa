"""Pins the CPU oracle against the committed golden vectors (tests/golden/make_golden.py):
independent torch.autograd Jacobians, SciPy loss tables and SciPy least_squares minima."""

import os

import numpy as np
import pytest

from conftest import GOLDEN, load_scene
from oracle import cpu_oracle as O


def test_loss_table():
    z = np.load(os.path.join(GOLDEN, "loss_table.npz"))
    for name, kind in (("soft_l1", 1), ("cauchy", 2)):
        for i, a in enumerate(z["a"]):
            for j, s in enumerate(z["s"]):
                r0, r1 = O.loss(kind, float(a), float(s))
                # 2a^2(sqrt(1+s/a^2)-1) cancels for s << a^2 in both implementations
                assert r0 == pytest.approx(z[f"{name}_rho0"][i, j], rel=1e-9, abs=4e-16 * a * a)
                assert r1 == pytest.approx(z[f"{name}_rho1"][i, j], rel=1e-12)
    assert O.loss(0, 1.0, 3.5) == (3.5, 1.0)


def test_block_jacobians_match_autograd():
    z = np.load(os.path.join(GOLDEN, "jacobians.npz"))
    for i in range(z["q"].shape[0]):
        r, Jc, Jp = O.reproj_block(z["q"][i], z["t"][i], z["K"][i], z["X"][i], z["xy"][i])
        np.testing.assert_allclose(r, z["r_reproj"][i], rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(Jc, z["Jc_reproj"][i], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(Jp, z["Jp_reproj"][i], rtol=1e-9, atol=1e-8)
        ok, r, Jc, Jp = O.depth_block(z["q"][i], z["t"][i], z["X"][i], z["d"][i])
        assert ok == 1
        np.testing.assert_allclose(r, z["r_depth"][i], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Jc, z["Jc_depth"][i], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(Jp, z["Jp_depth"][i], rtol=1e-9, atol=1e-11)


def test_depth_block_rejects_points_behind_camera():
    ok, *_ = O.depth_block([0, 0, 0, 1.0], [0, 0, -5.0], [0, 0, 1.0], 2.0)
    assert ok == 0


@pytest.mark.parametrize("name", ["scene_2x20", "scene_5x200", "scene_4x120_reproj"])
def test_converges_to_scipy_minimum(name):
    prob, z = load_scene(name)
    # tight tolerances: the Ceres-default ftol=1e-6 stops ~1e-5 above the minimum
    opt = O.default_options(function_tolerance=1e-14, parameter_tolerance=1e-14, max_num_iterations=400)
    s = O.solve(prob, opt)
    ref = float(z["scipy_cost"])
    assert s["final_cost"] == pytest.approx(ref, rel=1e-8)
    # same minimiser, not just same value
    np.testing.assert_allclose(prob.pts, z["scipy_pts"], atol=2e-5)
    np.testing.assert_allclose(prob.cam_t, z["scipy_cam_t"], atol=2e-5)
    dots = np.abs(np.sum(prob.cam_quat * z["scipy_cam_quat"], axis=1))
    np.testing.assert_allclose(dots, 1.0, atol=1e-9)


@pytest.mark.parametrize("name", ["scene_2x20", "scene_5x200", "scene_4x120_reproj"])
def test_default_options_within_north_star_tolerance(name):
    """With Ceres default tolerances the stopping point is within 1e-4 relative of the minimum."""
    prob, z = load_scene(name)
    s = O.solve(prob)
    ref = float(z["scipy_cost"])
    assert s["final_cost"] >= ref * (1 - 1e-9)
    assert (s["final_cost"] - ref) / ref < 1e-4
    assert s["termination"] in ("function_tolerance", "parameter_tolerance")
    # monotone accepted costs, radius grows on good steps
    tc = s["trace_cost"]
    assert all(tc[i + 1] <= tc[i] * (1 + 1e-12) for i in range(len(tc) - 1))


def test_cost_split_matches_trace():
    prob, _ = load_scene("scene_5x200")
    cr, cd = O.eval_cost(prob)
    s = O.solve(prob.copy())
    assert cr + cd == pytest.approx(s["initial_cost"], rel=1e-12)


@pytest.mark.parametrize("name", ["scene_2x20", "scene_5x200", "scene_4x120_reproj"])
def test_lm_trajectory_matches_independent_dense_implementation(name):
    """The oracle eliminates landmarks block by block (Schur complement); tests/golden/make_golden_lm.py runs the same
    published algorithm (Ceres 2.1 trust-region LM with default options) on the full dense Jacobian from torch
    forward-mode differentiation and numpy.linalg.solve.  Every iteration must agree: cost, radius, acceptance,
    iteration count, termination, final state."""
    import os

    g = np.load(os.path.join(GOLDEN, f"lm_trace_{name}.npz"), allow_pickle=False)
    prob = load_scene(name)[0]
    s = O.solve(prob)
    assert s["num_iterations"] == int(g["num_iterations"]) and s["termination"] == str(g["termination"])
    np.testing.assert_allclose(s["trace_cost"], g["trace_cost"], rtol=1e-11)
    np.testing.assert_allclose(s["trace_radius"], g["trace_radius"], rtol=1e-12)
    np.testing.assert_array_equal(s["trace_accepted"], g["trace_accepted"])
    np.testing.assert_allclose(prob.cam_quat, g["cam_quat"], atol=1e-11)
    np.testing.assert_allclose(prob.cam_t, g["cam_t"], atol=1e-11)
    np.testing.assert_allclose(prob.pts, g["pts"], atol=1e-10)


def test_point_covariances_match_autograd_golden():
    import os

    g = np.load(os.path.join(GOLDEN, "point_covs_scene_4x120_reproj.npz"))["covs"]
    prob = load_scene("scene_4x120_reproj")[0]
    c = O.point_covs(prob)
    np.testing.assert_allclose(c, g, rtol=1e-9, atol=1e-12 * np.abs(g).max())
