"""The device-side table build (csrc/build_dev.hip) against the host-thread build it replaces (`build()` in csrc/ba_solver.hip):
every table of two handles created from the same problem — one with MPSFM_DEV_BUILD=0 — compared bit for bit (rec_d = log depth:
within 2 ulp, the device's log is not libm's), then the solves compared."""

import ctypes as C
import os

import numpy as np
import pytest

from mpsfm_amd import capi
from mpsfm_amd.synthetic import local_window, make_scene

pytestmark = pytest.mark.gpu

TABLES = {0: ("chunks", np.int32), 1: ("chunk_cams", np.int32), 2: ("rec_cam", np.int32), 3: ("rec_pt", np.int32), 4: ("rec_meta", np.uint32),
          5: ("rec_xy", np.float64), 6: ("rec_d", np.float64), 7: ("rec_m", np.float64), 8: ("rec_a", np.float64), 9: ("pt_rec_start", np.int32),
          10: ("pt_kv", np.uint16), 11: ("fx_cam", np.int32), 12: ("fx_pt", np.int32), 13: ("fx_meta", np.uint32), 14: ("fx_xy", np.float64),
          15: ("fx_d", np.float64), 16: ("fx_m", np.float64), 17: ("fx_a", np.float64), 18: ("order", np.int32), 19: ("red_dests", np.int32),
          20: ("red_srcs", np.int32), 21: ("cam_slot", np.int32), 22: ("built_on_device", np.uint8), 23: ("blk_desc", np.uint32),
          24: ("blk_ent_start", np.int32), 25: ("ents", np.uint32)}


def tables(h):
    L = capi.lib()
    L.mpsfm_debug_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.mpsfm_debug_table.restype = C.c_int64
    out = {}
    for which, (name, dt) in TABLES.items():
        n = L.mpsfm_debug_table(h._h, which, None, 0)
        assert n >= 0, name
        buf = np.zeros(max(n, 1), np.uint8)
        assert L.mpsfm_debug_table(h._h, which, buf.ctypes.data, n) == n
        out[name] = buf[:n].view(dt).copy()
    return out


def both(prob, monkeypatch):
    monkeypatch.setenv("MPSFM_DEV_BUILD", "0")
    with capi.BAHandle(prob.copy()) as hh:
        th = tables(hh)
        hh.reset_state()
        sh = hh.solve()
    monkeypatch.delenv("MPSFM_DEV_BUILD")
    monkeypatch.setenv("MPSFM_SLAB_TABLES_HOST", "0")  # the slab reduction tables from the device kernels whatever the size
    with capi.BAHandle(prob.copy()) as hd:
        td = tables(hd)
        hd.reset_state()
        sd = hd.solve()
    monkeypatch.delenv("MPSFM_SLAB_TABLES_HOST")
    return th, td, sh, sd


def assert_same_tables(th, td):
    assert th["built_on_device"][0] == 0 and td["built_on_device"][0] == 1
    for name in th:
        if name == "built_on_device":
            continue
        a, b = th[name], td[name]
        assert a.shape == b.shape, name
        if name in ("rec_d", "fx_d"):
            assert np.all(np.abs(a - b) <= 4 * np.spacing(np.abs(a))), name
        else:
            np.testing.assert_array_equal(a, b, err_msg=name)


@pytest.mark.parametrize("ncam,npts,depth,seed", [(50, 20000, False, 0), (24, 9000, True, 3), (60, 30000, True, 4), (6, 300, True, 5),
                                                  (130, 12000, True, 6),    # camera sets of three 64-bit words
                                                  (600, 9000, False, 7)])   # more than 512 slots: the wave-per-segment cut
def test_device_build_equals_host_build(monkeypatch, ncam, npts, depth, seed):
    prob, _ = make_scene(ncam, npts, depth, seed=seed)
    th, td, sh, sd = both(prob, monkeypatch)
    assert_same_tables(th, td)
    assert len(th["chunks"]) > 0 and len(th["red_srcs"]) > 0
    assert sh["num_iterations"] == sd["num_iterations"] and sh["termination"] == sd["termination"]
    assert sd["final_cost"] == pytest.approx(sh["final_cost"], rel=1e-10)


def test_device_build_with_constant_cameras_points_and_fixed_blocks(monkeypatch):
    """A local window: constant outside cameras (records with lcam 255), constant landmarks, blocks with both constant (the fixed
    list), cameras without any block, landmarks without any block."""
    base, _ = make_scene(30, 6000, True, seed=11)
    prob = local_window(base, list(range(8, 14)), ref_cam=10)[0]
    prob.pt_const[::7] = 1
    prob.pose_const[-1] = 0  # a variable camera that may have no block at all
    th, td, sh, sd = both(prob, monkeypatch)
    assert_same_tables(th, td)
    assert len(th["fx_cam"]) > 0 and (th["rec_meta"] & 0xff == 255).any()
    assert sd["final_cost"] == pytest.approx(sh["final_cost"], rel=1e-10) and sh["num_iterations"] == sd["num_iterations"]


def test_general_chunks_get_their_pair_tables_from_the_host(monkeypatch):
    """Landmarks with more than 16 cameras (and landmarks with two records of one camera) form general chunks behind the dense ones:
    the device build keeps everything else and asks the host for their pair tables only — all tables equal the host build's."""
    prob, _ = make_scene(40, 3000, True, seed=2, max_track=40, track_mean=12.0)
    dup = np.flatnonzero(prob.obs_pt == 5)[:1]   # a second reprojection block of one camera on landmark 5
    prob.obs_cam = np.concatenate([prob.obs_cam, prob.obs_cam[dup]]); prob.obs_pt = np.concatenate([prob.obs_pt, prob.obs_pt[dup]])
    prob.obs_xy = np.concatenate([prob.obs_xy, prob.obs_xy[dup] + 0.5])
    th, td, sh, sd = both(prob, monkeypatch)
    assert_same_tables(th, td)
    dense = th["chunks"].reshape(-1, 12)[:, 10]
    assert (dense == 0).any() and (dense == 1).any() and len(th["ents"]) > 0
    assert sd["final_cost"] == pytest.approx(sh["final_cost"], rel=1e-10) and sh["num_iterations"] == sd["num_iterations"]


def test_long_tracks_take_the_host_build():
    """A landmark with more blocks than a chunk holds (here: seen by 300 cameras) is swept by a workgroup of its own; such problems
    keep the host phases, the handle says so and solves."""
    prob, truth = make_scene(300, 2000, False, seed=4)
    from mpsfm_amd.synthetic import R_from_quat
    R = R_from_quat(truth["cam_quat"])
    Xc = R @ truth["pts"][0] + truth["cam_t"]
    uv = np.stack([1200 * Xc[:, 0] / Xc[:, 2] + 800, 1200 * Xc[:, 1] / Xc[:, 2] + 600], 1)
    prob.obs_cam = np.concatenate([prob.obs_cam, np.arange(300, dtype=np.int32)]); prob.obs_pt = np.concatenate([prob.obs_pt, np.zeros(300, np.int32)])
    prob.obs_xy = np.concatenate([prob.obs_xy, uv])
    with capi.BAHandle(prob.copy()) as h:
        t = tables(h)
        assert t["built_on_device"][0] == 0
        s = h.solve()
    assert s["final_cost"] < s["initial_cost"]
