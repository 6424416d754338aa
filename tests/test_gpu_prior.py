"""Row f3: the depth-block gather kernel (mpsfm_depth_blocks) against its NumPy restatement (oracle/prior_oracle.py) and
directly against vectors computed by the reference's own PriorUtils (tests/golden/reference_priorutils.npz)."""

import os
import time

import numpy as np
import pytest

from backends import OracleBackend
from conftest import GOLDEN
from mpsfm_amd import capi
from mpsfm_amd.sfm.mapper.bundle_adjustment import Optimizer
from numpy_scene import scene_from_problem
from mpsfm_amd.sfm.scene.prior_gather import gather_bundle
from mpsfm_amd.synthetic import make_scene
from oracle import prior_oracle

pytestmark = pytest.mark.gpu


def test_sampling_equals_the_references_own_priorutils_outputs():
    """The kernel's bilinear samples of depth and validity against PriorUtils.data_at_kps / valid_at_kps outputs that the
    reference's own code produced (fixture): values to 1e-13, the == 1 validity decision exactly."""
    z = np.load(os.path.join(GOLDEN, "reference_priorutils.npz"))
    for tag in ("a", "b", "c"):
        kps = z[f"pu_{tag}_kps"]
        n = len(kps)
        out = capi.depth_blocks([z[f"pu_{tag}_data"]], [z[f"pu_{tag}_valid"]], [z[f"pu_{tag}_s"][0]], [z[f"pu_{tag}_s"][1]],
                                [[0, 0, 0, 1.0]], [[0, 0, 5.0]], np.zeros(n, np.int32), kps, np.full(n, 0.01), np.zeros(n, np.int32),
                                np.zeros((1, 3)))
        np.testing.assert_allclose(out["depth"], z[f"pu_{tag}_data_at_kps"], rtol=1e-13, atol=1e-13)
        np.testing.assert_array_equal((out["flags"] & 1) != 0, z[f"pu_{tag}_valid_at_kps"])
        np.testing.assert_allclose(out["depth3d"], 5.0)


def test_gather_kernel_matches_numpy_restatement():
    prob, truth = make_scene(9, 2500, True, seed=61)
    sc = scene_from_problem(prob, truth, seed=6)
    rng = np.random.default_rng(1)
    ids = sorted(sc.images)
    # maps of different sizes, non-positive depths, keypoints off the map
    im = sc.images[ids[2]]
    im.depth.data = im.depth.data[:-7, :-5].copy()
    im.depth.valid = im.depth.valid[:-7, :-5].copy()
    sc.images[ids[3]].depth.data[::6, ::4] = -1.0
    sc.images[ids[4]].kps[::9] += rng.uniform(-900, 900, sc.images[ids[4]].kps[::9].shape)
    g = gather_bundle(sc, ids, "update")
    assert g["obs_img"].max() == len(ids) - 1 and len(g["obs_img"]) == prob.n_obs
    kw = dict(depth_maps=g["depth_maps"], valid_maps=g["valid_maps"], sx=g["sx"], sy=g["sy"], cam_quat=g["cam_quat"], cam_t=g["cam_t"],
              obs_img=g["obs_img"], obs_xy=g["obs_xy"], obs_var=g["obs_var"], obs_pt=g["obs_pt"], pts=sc.point3D_coordinates(g["point_ids"]),
              scale_filter_factor=1.5, multiplier=0.25 * 3.7 * 2)
    hip, ref = capi.depth_blocks(**kw), prior_oracle.depth_blocks(**kw)
    np.testing.assert_array_equal(hip["flags"], ref["flags"])  # mask decisions: exact
    assert len(np.unique(hip["flags"])) >= 6
    np.testing.assert_allclose(hip["depth"], ref["depth"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(hip["depth3d"], ref["depth3d"], rtol=1e-13)
    ok = (hip["flags"] & 3) == 3
    for k in ("magnitude", "param", "whitened"):
        np.testing.assert_allclose(hip[k][ok], ref[k][ok], rtol=1e-12, atol=1e-12)


def test_ba_through_the_gather_equals_oracle_backend_and_is_fast():
    """Optimizer.ba() end to end on 40 cameras / 30 k landmarks: same flat problem as with the NumPy restatement, and the
    wall time of one call (assembly + one-shot solve + write-back)."""
    prob, truth = make_scene(40, 30000, True, seed=5)
    sc_g, sc_o = scene_from_problem(prob, truth, seed=1), scene_from_problem(prob, truth, seed=1)
    og, oo = Optimizer({}, sc_g, None), Optimizer({}, sc_o, None, backend=OracleBackend())
    b = {"optim_ids": set(sc_g.images), "pts3D": set(sc_g.points3D), "constpoints": set()}
    fg, _ = og._build_problem(b, False, True, mode="global", allow_scale_filter=True, solve=False)
    fo, _ = oo._build_problem(b, False, True, mode="global", allow_scale_filter=True, solve=False)
    for k in ("dobs_cam", "dobs_pt"):
        np.testing.assert_array_equal(getattr(fg.prob, k), getattr(fo.prob, k))
    for k in ("dobs_depth", "dobs_magnitude", "dobs_param"):
        np.testing.assert_allclose(getattr(fg.prob, k), getattr(fo.prob, k), rtol=1e-12)
    og.update_truncation_multiplier(list(sc_g.images))
    oo.update_truncation_multiplier(list(sc_o.images))
    assert og.truncation_multiplier == pytest.approx(oo.truncation_multiplier, rel=1e-12)
    og.ba(b, mode="global", allow_scale_filter=True)  # warm-up (allocator cache, code objects)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        r, _ = og.ba(b, mode="global", allow_scale_filter=True)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"Optimizer.ba() 40 cameras / 30 k landmarks: {min(ts):.1f} ms wall ({r.summary['num_iterations']} LM iterations, "
          f"solve {1e3 * r.summary['time_total_s']:.1f} ms)")
