"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol and refuses to
compute without a GPU (no silent fallback)."""

import ctypes
import os
import re

import pytest

from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mpsfm_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(mpsfm_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/mpsfm_hip.h but not exported"
    assert set(capi.EXPORTS) <= declared
    assert L.mpsfm_abi_version() == 2


def test_triangulator_and_gather_entry_points_check_arguments_first():
    import ctypes as C

    from mpsfm_amd.sfm.mapper.track_engine import CTriOptions

    L = capi.lib()
    o = CTriOptions()
    L.mpsfm_tri_default_options(C.byref(o))
    assert (o.max_transitivity, o.create_max_angle_error, o.merge_max_reproj_error, o.complete_max_transitivity) == (1, 2.0, 4.0, 5)
    assert (o.re_max_angle_error, o.re_min_ratio, o.re_max_trials, o.min_angle, o.ignore_two_view_tracks) == (5.0, 0.2, 1, 1.5, 1)
    L.mpsfm_triangulator_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    h = C.c_void_p(None)
    assert L.mpsfm_triangulator_create(None, 0, C.byref(h)) == -1
    L.mpsfm_triangulator_set_state.argtypes = [C.c_void_p, C.c_void_p]
    assert L.mpsfm_triangulator_set_state(None, None) == -1
    L.mpsfm_depth_blocks.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
    assert L.mpsfm_depth_blocks(None, 0, None, None, None, None, None, None) == -1


def test_default_options_are_ceres_defaults():
    o = capi.default_options()
    assert o.max_num_iterations == 50
    assert o.function_tolerance == 1e-6 and o.gradient_tolerance == 1e-10 and o.parameter_tolerance == 1e-8
    assert o.initial_trust_region_radius == 1e4 and o.min_relative_decrease == 1e-3
    assert o.min_lm_diagonal == 1e-6 and o.max_lm_diagonal == 1e32 and o.jacobi_scaling == 1


def test_no_device_means_loud_failure_not_fallback():
    if capi.device_count() > 0:
        pytest.skip("a gfx950 device is visible")
    prob, _ = make_scene(3, 30, True, seed=0)
    with pytest.raises(capi.MpsfmHipError) as e:
        capi.ba_solve(prob)
    assert e.value.code == -2
    with pytest.raises(capi.MpsfmHipError):
        capi.point_covs(prob)


def test_invalid_problem_is_rejected_before_touching_the_device():
    prob, _ = make_scene(3, 30, True, seed=0)
    prob.obs_cam[0] = 99  # bypasses BAProblem.validate on purpose
    with pytest.raises(capi.MpsfmHipError) as e:
        capi.ba_solve(prob)
    assert e.value.code == -1


def test_integration_entry_points_validate_arguments_first():
    """Argument errors of the depth-integration entry points are reported as MPSFM_EINVAL before any device is
    touched; an empty batch is a no-op; without a device the calls fail loudly (MPSFM_ENODEVICE)."""
    import ctypes as C

    import numpy as np

    from mpsfm_amd.synthetic_maps import make_maps

    L = capi.lib()
    L.mpsfm_integrate_depth_batch.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    assert L.mpsfm_integrate_depth_batch(0, None, 0, None, None) == 0
    assert L.mpsfm_integrate_depth_batch(-1, None, 0, None, None) == -1
    assert L.mpsfm_integrate_depth_batch(2, None, 0, None, None) == -1
    assert capi.integrate_depth_batch([]) == []
    m = make_maps(12, 16, seed=0, n_sparse=5)
    nu = m["normals_uncertainty"]
    nvar = np.stack([nu[..., 0, 0], nu[..., 1, 1], nu[..., 2, 2]], -1)
    args = (m["depth_prior"], m["depth_uncertainty"], m["valid"], m["normals"], nvar, m["depth_init"], m["K"])
    with pytest.raises(capi.MpsfmHipError) as e:
        capi.integration_variances(*args, np.array([[1, 1]]), rtol=0.0)
    assert e.value.code == -1
    if capi.device_count() == 0:
        with pytest.raises(capi.MpsfmHipError) as e:
            capi.integration_variances(*args, np.array([[1, 1]]))
        assert e.value.code == -2
        with pytest.raises(capi.MpsfmHipError) as e:
            capi.integrate_depth(*args, m["kps"], m["depth3d"], m["zvars3d"])
        assert e.value.code == -2


def test_null_arrays_are_einval_not_a_crash():
    """mpsfm_point_covs / mpsfm_triangulate_tracks / mpsfm_filter_tracks check their pointers like
    mpsfm_ba_solve does (a C caller passing NULL gets MPSFM_EINVAL, not a segfault)."""
    import ctypes as C

    import numpy as np

    from mpsfm_amd.problem import CProblem, CState, CTracks

    L = capi.lib()
    prob, _ = make_scene(3, 30, False, seed=0)
    covs = np.zeros((prob.n_pts, 3, 3))
    for field in ("obs_cam", "obs_pt", "obs_xy", "cam_intr_idx", "cam_intr"):
        cp, cs = prob.c_problem(), prob.c_state()
        setattr(cp, field, None)
        assert L.mpsfm_point_covs(C.byref(cp), C.byref(cs), 0, covs.ctypes.data) == -1, field
    for field in ("cam_quat_xyzw", "cam_t", "pts"):
        cp, cs = prob.c_problem(), prob.c_state()
        setattr(cs, field, None)
        assert L.mpsfm_point_covs(C.byref(cp), C.byref(cs), 0, covs.ctypes.data) == -1, field
    assert L.mpsfm_point_covs(None, None, 0, None) == -1
    start = np.array([0, 2, 4], np.int64)
    el_cam = np.array([0, 1, 1, 2], np.int32)
    el_xy = np.zeros((4, 2))
    xyz = np.zeros((2, 3))

    def tracks():
        t = CTracks()
        t.n_cams, t.n_tracks, t.n_intr = 3, 2, 1
        t.cam_quat_xyzw, t.cam_t = prob.cam_quat.ctypes.data, prob.cam_t.ctypes.data
        t.cam_intr, t.cam_intr_idx = prob.cam_intr.ctypes.data, prob.cam_intr_idx.ctypes.data
        t.track_start, t.el_cam, t.el_xy = start.ctypes.data, el_cam.ctypes.data, el_xy.ctypes.data
        return t

    for field in ("cam_quat_xyzw", "cam_t", "cam_intr", "cam_intr_idx", "track_start", "el_cam", "el_xy"):
        t = tracks()
        setattr(t, field, None)
        assert L.mpsfm_triangulate_tracks(C.byref(t), 0, xyz.ctypes.data) == -1, field
        assert L.mpsfm_filter_tracks(C.byref(t), xyz.ctypes.data, 0, None, None, None) == -1, field
    t = tracks()
    assert L.mpsfm_triangulate_tracks(C.byref(t), 0, None) == -1


def test_non_pinhole_camera_models_are_refused():
    import types

    import numpy as np

    from mpsfm_amd.sfm.mapper.bundle_adjustment import pinhole_params

    cam = types.SimpleNamespace(params=np.array([1200.0, 1190.0, 800.0, 600.0]))
    np.testing.assert_array_equal(pinhole_params(cam), cam.params)
    cam.model = types.SimpleNamespace(name="PINHOLE")
    np.testing.assert_array_equal(pinhole_params(cam), cam.params)
    simple = types.SimpleNamespace(params=np.array([1000.0, 800.0, 600.0]), model=types.SimpleNamespace(name="SIMPLE_PINHOLE"))
    np.testing.assert_array_equal(pinhole_params(simple), [1000.0, 1000.0, 800.0, 600.0])
    for name in ("SIMPLE_RADIAL", "OPENCV"):
        with pytest.raises(NotImplementedError):
            pinhole_params(types.SimpleNamespace(params=np.zeros(8), model=types.SimpleNamespace(name=name)))
