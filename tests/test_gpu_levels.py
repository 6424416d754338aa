"""Level-scheduled factorisation of the reduced camera system (dense_chol.hip: k_chol_level, chol_plan.hip): whatever the
slot order — the caller's, nested dissection of any depth, the skyline fall-back without a camera graph — the dense
solution equals numpy's on the SAME reduced system (which the accessors return in the caller's camera order), and the
whole solve follows the oracle's trajectory."""

import numpy as np
import pytest

from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene, shuffle_cameras
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu

VARIANTS = {
    "auto": {},
    "callers_order": {"MPSFM_CHOL_ND": "-1"},
    "rcm_only": {"MPSFM_CHOL_ND": "0"},
    "depth1": {"MPSFM_CHOL_ND": "1"},
    "depth3": {"MPSFM_CHOL_ND": "3"},
    "no_graph_skyline": {"MPSFM_CHOL_GRAPH": "0"},
    "depth2_backward_by_levels": {"MPSFM_CHOL_ND": "2", "MPSFM_CHOL_INVERSE": "0"},
    # not a slot order: every chunk forms its Schur blocks from the pair lists (the default does so only for chunks of more
    # than eight cameras, the others take the dense product on the matrix pipe) — the same S and rhs either way
    "sweep_pair_lists": {"MPSFM_SWEEP_DENSE": "0"},
}


@pytest.fixture(scope="module")
def orbit():
    prob, _ = make_scene(150, 12000, True, seed=31)
    ref = O.reduced_system(prob, radius=1e3)
    sol = O.solve(prob.copy())
    return prob, ref, sol


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_every_slot_order_solves_the_same_system(variant, orbit, monkeypatch):
    prob, ref, so = orbit
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    with capi.BAHandle(prob.copy()) as h:
        assert h.reduced_dim == 6 * (prob.n_cams - 1)
        plan = h.dense_plan()
        assert plan["slots"] == prob.n_cams - 1 and plan["tile_columns"] >= (6 * plan["slots"] + 31) // 32  # padding columns align the parts
        assert 1 <= plan["levels"] <= plan["tile_columns"]
        if variant in ("callers_order", "no_graph_skyline"):
            assert plan["nd_depth"] == -1 and plan["levels"] == plan["tile_columns"] == (6 * plan["slots"] + 31) // 32
        if variant in ("auto", "depth1", "depth3"):
            assert plan["nd_depth"] >= 1 and plan["levels"] < plan["tile_columns"]  # the orbit's chain is cut into shorter ones
        assert plan["inverse_accumulators"] == (0 if variant == "depth2_backward_by_levels" else 1)
        h.sweep_once(1e3)
        S, rhs = h.reduced_system()
        np.testing.assert_allclose(S, ref["S"], rtol=0, atol=1e-11 * np.abs(ref["S"]).max())
        np.testing.assert_allclose(rhs, ref["rhs"], rtol=0, atol=1e-11 * np.abs(ref["rhs"]).max())
        h.dense_solve_once()
        y = h.dense_solution()
        y_np = np.linalg.solve(S, rhs)
        np.testing.assert_allclose(y, y_np, rtol=0, atol=1e-9 * np.abs(y_np).max())
    pg = prob.copy()
    sg = capi.ba_solve(pg)
    assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"]
    assert sg["trace_accepted"] == so["trace_accepted"]
    np.testing.assert_allclose(sg["trace_cost"], so["trace_cost"], rtol=1e-9)


def test_shuffled_cameras_and_two_scenes_in_one_problem():
    """An unordered camera list is re-ordered by the camera graph (the result does not depend on the caller's order beyond
    rounding); two scenes that share nothing become independent chains of one factorisation."""
    prob, _ = make_scene(130, 9000, True, seed=44)
    shuf, perm = shuffle_cameras(prob, seed=3)
    s0, s1 = capi.ba_solve(p0 := prob.copy()), capi.ba_solve(p1 := shuf.copy())
    assert s0["num_iterations"] == s1["num_iterations"] and s0["termination"] == s1["termination"]
    assert s1["final_cost"] == pytest.approx(s0["final_cost"], rel=1e-10)
    np.testing.assert_allclose(p1.cam_t, p0.cam_t[perm], atol=1e-8)
    np.testing.assert_allclose(p1.pts, p0.pts, atol=1e-8)
    so = O.solve(shuf.copy())
    assert s1["num_iterations"] == so["num_iterations"] and s1["final_cost"] == pytest.approx(so["final_cost"], rel=1e-9)

    a, _ = make_scene(40, 3000, True, seed=7)
    b, _ = make_scene(55, 4000, True, seed=8)
    from mpsfm_amd.problem import BAProblem
    both = BAProblem(
        cam_quat=np.vstack([a.cam_quat, b.cam_quat]), cam_t=np.vstack([a.cam_t, b.cam_t]), pts=np.vstack([a.pts, b.pts]),
        cam_intr=a.cam_intr, cam_intr_idx=np.concatenate([a.cam_intr_idx, b.cam_intr_idx]),
        pose_const=np.concatenate([a.pose_const, b.pose_const]), pt_const=np.concatenate([a.pt_const, b.pt_const]),
        obs_cam=np.concatenate([a.obs_cam, b.obs_cam + a.n_cams]), obs_pt=np.concatenate([a.obs_pt, b.obs_pt + a.n_pts]),
        obs_xy=np.vstack([a.obs_xy, b.obs_xy]), gauge_axis_cam=a.gauge_axis_cam,
        dobs_cam=np.concatenate([a.dobs_cam, b.dobs_cam + a.n_cams]), dobs_pt=np.concatenate([a.dobs_pt, b.dobs_pt + a.n_pts]),
        dobs_depth=np.concatenate([a.dobs_depth, b.dobs_depth]), dobs_magnitude=np.concatenate([a.dobs_magnitude, b.dobs_magnitude]),
        dobs_param=np.concatenate([a.dobs_param, b.dobs_param]))
    sg, so = capi.ba_solve(pg := both.copy()), O.solve(po := both.copy())
    assert sg["num_iterations"] == so["num_iterations"] and sg["termination"] == so["termination"]
    assert sg["final_cost"] == pytest.approx(so["final_cost"], rel=1e-9)
    np.testing.assert_allclose(pg.cam_t, po.cam_t, atol=1e-6)
