"""GPU tests at the real drop-in seam: the ObservationManager-shaped adaptor and the
triangulator's depth lifting on the HIP kernels, and the mapper's call sequences (adjust_bundle, _refinement,
post_init_refinement, post_registration_refinement, iterative_local_refinement) replayed with the HIP backend against
the same sequence with the oracle backend."""

import copy

import numpy as np
import pytest

from mapper_replay import MapperReplay
from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from mpsfm_amd.sfm.mapper.triangulator import MpsfmTriangulator, track_quality
from numpy_scene import ObservationManager, scene_from_problem
from mpsfm_amd.sfm.scene.observations import HipObservationManager
from mpsfm_amd.synthetic import make_scene
from numpy_integrable import NumpyIntegrableImage, NumpyNormals
from oracle import cpu_oracle as O
from test_observations_cpu import assert_same_state, make_dirty_scene, oracle_numerics

pytestmark = pytest.mark.gpu


from backends import OracleBackend  # noqa: E402


def _oracle_side(sc):
    """A deep copy of the scene whose filters use the oracle's numerics and whose solver is the CPU oracle."""
    b = copy.deepcopy(sc)
    b.obs = HipObservationManager(b, ObservationManager(b), numerics=oracle_numerics)
    return b


def test_observation_filters_on_hip_match_oracle_decisions_exactly():
    """a-12: every filter of the adaptor, HIP numerics vs oracle/tri_oracle.c numerics: identical return values and
    identical scenes afterwards (decisions are booleans: exact)."""
    a = make_dirty_scene(11, n_cams=10, n_pts=1500)
    b = _oracle_side(a)
    assert type(a.obs) is HipObservationManager and a.obs._numerics is not oracle_numerics
    ids = sorted(a.points3D)
    np.testing.assert_array_equal(a.obs.find_small_angle_points_mask(1.5, ids), b.obs.find_small_angle_points_mask(1.5, ids))
    assert a.obs.filter_observations_with_negative_depth() == b.obs.filter_observations_with_negative_depth()
    assert_same_state(a, b)
    some = set(ids[::3])
    na, nb = a.obs.filter_points3D(4.0, 1.5, some), b.obs.filter_points3D(4.0, 1.5, some)
    assert na == nb and na > 50
    assert_same_state(a, b)
    na, nb = a.obs.filter_all_points3D(4.0, 0.001), b.obs.filter_all_points3D(4.0, 0.001)
    assert na == nb and na > 50
    assert_same_state(a, b)
    assert 100 < len(a.points3D) < 1500


def test_triangulator_lifts_through_the_hip_kernel():
    """a-11: lift_low_parallax takes its mask from mpsfm_filter_tracks itself; same points lifted, same coordinates as
    with the oracle's angles."""
    prob, truth = make_scene(6, 300, True, seed=11)
    a = scene_from_problem(prob, truth, seed=1)
    b = _oracle_side(a)
    ids = sorted(a.points3D)
    ang, _, _ = track_quality(a, ids)
    ang_o, _, _ = oracle_numerics(*_tracks(b, ids))
    np.testing.assert_allclose(ang, ang_o, rtol=0, atol=1e-12)
    thr = float(np.rad2deg(np.median(ang)))
    new_a = MpsfmTriangulator({"colmap_options": {}}, a, None).lift_low_parallax(ids, thr)
    # oracle side: the same rule with the oracle's angles
    tri_b = MpsfmTriangulator({"colmap_options": {}}, b, None)
    new_b = tri_b._lift_points(np.array(ids)[ang_o < np.deg2rad(thr)])
    assert new_a == new_b and 0 < len(new_a)
    assert_same_state(a, b)
    for pid in new_a:
        np.testing.assert_array_equal(a.points3D[pid].xyz, b.points3D[pid].xyz)


def _tracks(sc, ids):
    from mpsfm_amd.sfm.mapper.triangulator import tracks_from_scene

    tr, _ = tracks_from_scene(sc, ids)
    return tr, sc.point3D_coordinates(ids), 0


def _normals_for(sc, imid, seed):
    H, W = sc.images[imid].depth.data_prior.shape
    rng = np.random.default_rng(seed)
    nrm = rng.normal(0, 0.05, (H, W, 3)) + np.array([0.0, 0.0, -1.0])
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    ncov = np.zeros((H, W, 3, 3))
    ncov[..., 0, 0] = ncov[..., 1, 1] = ncov[..., 2, 2] = 0.05**2
    return NumpyNormals(nrm, ncov)


class _CopyIntegration:
    """Oracle side of the replay: takes the integrated map the HIP side computed for the same image (the integration
    solve has its own parity tests in test_gpu_integration.py; here the BA path is what is compared)."""

    def __init__(self, src_image, dst_image):
        self.src, self.dst = src_image, dst_image

    def integrate(self, cache_device="cpu"):
        self.dst.depth.data = self.src.depth.data.copy()
        return True


def _replay_pair(n_cams, n_pts, seed, map_size=(64, 48)):
    prob, truth = make_scene(n_cams, n_pts, True, seed=seed)
    a = scene_from_problem(prob, truth, map_size=map_size, seed=seed + 1)
    b = _oracle_side(a)
    int_a = {i: NumpyIntegrableImage(a, i, _normals_for(a, i, 100 + i)) for i in a.images}
    int_b = {i: _CopyIntegration(a.images[i], b.images[i]) for i in b.images}
    ma = MapperReplay(a, Optimizer({}, a, None), None, int_a)
    mb = MapperReplay(b, Optimizer({}, b, None, backend=OracleBackend()), None, int_b)
    return a, b, ma, mb


def _assert_scenes_close(a, b, atol=2e-6):
    assert set(a.points3D) == set(b.points3D)
    for i in a.images:
        np.testing.assert_allclose(a.images[i].cam_from_world.translation, b.images[i].cam_from_world.translation, atol=atol)
        assert abs(a.images[i].cam_from_world.rotation.quat @ b.images[i].cam_from_world.rotation.quat) == pytest.approx(1, abs=1e-10)
    pa = np.array([a.points3D[p].xyz for p in sorted(a.points3D)])
    pb = np.array([b.points3D[p].xyz for p in sorted(b.points3D)])
    np.testing.assert_allclose(pa, pb, atol=atol)


def test_global_refinement_contract_replay():
    """a-13, reference mapper/base.py:420-440, 633-654 and the final pass of :403-406:
    integrate_bundle(all) -> update_truncation_multiplier -> ba(global, allow_scale_filter, param_multiplier=0.125,
    final=True) -> filter_bundle -> (complete/merge), HIP against the oracle backend on identical scenes."""
    a, b, ma, mb = _replay_pair(8, 700, 31)
    for m in (ma, mb):
        assert m.post_init_refinement()
    assert ma.calls == mb.calls and [c[0] for c in ma.calls[:3]] == ["calculate_point_covs", "optimize_prior_shiftscale", "refine_3d_points"]
    _assert_scenes_close(a, b)
    for m in (ma, mb):
        m.calls.clear()
        bundle = m.find_global_bundle()
        m.optimizer.calculate_point_covs(bundle)
        changed, ok = m._refinement(bundle, False, mode="global", allow_scale_filter=True, param_multiplier=0.125, final=True)
        assert ok and 0 <= changed < 0.5
    names = [c[0] for c in ma.calls]
    assert names[: len(a.images)] == ["integrate"] * len(a.images)
    assert names[len(a.images):] == ["update_truncation_multiplier", "ba", "filter_bundle"] and ma.calls == mb.calls
    assert ma.optimizer.truncation_multiplier == pytest.approx(mb.optimizer.truncation_multiplier, rel=1e-10)
    sa, sb = ma.optimizer.last_summary, mb.optimizer.last_summary
    assert sa["num_iterations"] == sb["num_iterations"] and sa["termination"] == sb["termination"]
    assert sa["final_cost"] == pytest.approx(sb["final_cost"], rel=1e-8)
    np.testing.assert_allclose(sa["trace_cost"], sb["trace_cost"], rtol=1e-9)
    _assert_scenes_close(a, b)
    assert_same_state_ids(a, b)
    for p in a.point_covs.data:
        np.testing.assert_allclose(a.point_covs.data[p], b.point_covs.data[p], rtol=1e-8, atol=1e-18)


def assert_same_state_ids(a, b):
    assert {p: sorted((e.image_id, e.point2D_idx) for e in q.track.elements) for p, q in a.points3D.items()} == \
           {p: sorted((e.image_id, e.point2D_idx) for e in q.track.elements) for p, q in b.points3D.items()}


def test_post_registration_and_local_refinement_contract_replay():
    """a-13, reference mapper/base.py:541-617 and :442-474 for a freshly 'registered' image: reset depth ->
    filter -> refine_3d_points -> filter -> calculate_point_covs(observed) -> optimize_prior_shiftscale(metric filter) ->
    rescale + activate -> integrate([imid]) -> refine_3d_points -> filter; then two local refinements
    (calculate_point_covs -> integrate(ref) -> ba(local) -> filter)."""
    a, b, ma, mb = _replay_pair(9, 900, 37)
    for m in (ma, mb):
        assert m.post_init_refinement()
    imid = sorted(a.images)[4]
    for m in (ma, mb):
        m.calls.clear()
        assert m.post_registration_refinement(imid)
    names = [c[0] for c in ma.calls]
    assert names == ["filter_bundle", "refine_3d_points", "filter_bundle", "calculate_point_covs", "optimize_prior_shiftscale",
                     "integrate", "refine_3d_points", "filter_bundle"] and ma.calls == mb.calls
    assert a.images[imid].depth.scale == pytest.approx(b.images[imid].depth.scale, rel=1e-10)
    _assert_scenes_close(a, b)
    assert_same_state_ids(a, b)
    for m in (ma, mb):
        m.calls.clear()
        assert m.iterative_local_refinement(imid)
    assert ma.calls == mb.calls and ("ba", "local") in ma.calls
    sa, sb = ma.optimizer.last_summary, mb.optimizer.last_summary
    assert sa["num_iterations"] == sb["num_iterations"] and sa["final_cost"] == pytest.approx(sb["final_cost"], rel=1e-8)
    # a 4-iteration local BA stops on the function tolerance with the camera depth direction barely constrained:
    # equal costs to 1e-8, states to 1e-5 along that direction (6e-6 observed)
    _assert_scenes_close(a, b, atol=3e-5)
    assert_same_state_ids(a, b)
