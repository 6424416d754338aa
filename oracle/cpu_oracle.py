"""ctypes wrapper of oracle/liboracle_ba.so — TEST INFRASTRUCTURE, not product code.

The library is the double-precision CPU restatement of the reference's pyceres/pycolmap path
(see the header of oracle/ba_oracle.c for what it follows and why parity is unpinned).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from mpsfm_amd.problem import ALLREDUCE_FN, BAProblem, COptions, CSummary, Tracks

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle_ba.so")
    srcs = [os.path.join(_HERE, f) for f in ("ba_oracle.c", "tri_oracle.c", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "mpsfm_hip.h"))
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_loss.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p]
        _LIB.oracle_loss.restype = None
        _LIB.oracle_depth_block.argtypes = [C.c_void_p] * 3 + [C.c_double] * 3 + [C.c_void_p] * 3
    return _LIB


def default_options(**kw) -> COptions:
    o = COptions()
    lib().oracle_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def solve(prob: BAProblem, options: COptions | None = None) -> dict:
    """Solves in place (prob.cam_quat / cam_t / pts are updated) and returns the summary."""
    o = options if options is not None else default_options()
    cp, cs, sm = prob.c_problem(), prob.c_state(), CSummary()
    rc = lib().oracle_ba_solve(C.byref(cp), C.byref(cs), C.byref(o), C.byref(sm))
    if rc != 0:
        raise RuntimeError(f"oracle_ba_solve failed with {rc}")
    return sm.to_dict()


def eval_cost(prob: BAProblem) -> tuple[float, float]:
    out = np.zeros(2)
    cp, cs = prob.c_problem(), prob.c_state()
    rc = lib().oracle_ba_eval_cost(C.byref(cp), C.byref(cs), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_ba_eval_cost failed with {rc}")
    return float(out[0]), float(out[1])


def reduced_dim(prob: BAProblem) -> int:
    cp = prob.c_problem()
    return int(lib().oracle_reduced_dim(C.byref(cp)))


def reduced_system(prob: BAProblem, radius: float = 1e4, jacobi: bool = True) -> dict:
    n = reduced_dim(prob)
    S = np.zeros((n, n))
    rhs = np.zeros(n)
    cam_scale = np.zeros((prob.n_cams, 6))
    pt_scale = np.zeros((prob.n_pts, 3))
    yc = np.zeros(n)
    yp = np.zeros((prob.n_pts, 3))
    mcc = C.c_double(0.0)
    cp, cs = prob.c_problem(), prob.c_state()
    f = lib().oracle_reduced_system
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int] + [C.c_void_p] * 7
    rc = f(
        C.byref(cp), C.byref(cs), radius, int(jacobi),
        S.ctypes.data, rhs.ctypes.data, cam_scale.ctypes.data, pt_scale.ctypes.data,
        yc.ctypes.data, yp.ctypes.data, C.addressof(mcc),
    )
    if rc < 0:
        raise RuntimeError(f"oracle_reduced_system failed with {rc}")
    return dict(S=S, rhs=rhs, cam_scale=cam_scale, pt_scale=pt_scale, yc=yc, yp=yp, model_cost_change=mcc.value)


def point_covs(prob: BAProblem) -> np.ndarray:
    covs = np.zeros((prob.n_pts, 3, 3))
    cp, cs = prob.c_problem(), prob.c_state()
    rc = lib().oracle_point_covs(C.byref(cp), C.byref(cs), covs.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_point_covs failed with {rc}")
    return covs


def loss(loss_type: int, a: float, s: float) -> tuple[float, float]:
    out = np.zeros(2)
    lib().oracle_loss(loss_type, a, s, out.ctypes.data_as(C.c_void_p))
    return float(out[0]), float(out[1])


def reproj_block(q, t, K, X, xy):
    q, t, K, X, xy = (np.ascontiguousarray(a, np.float64) for a in (q, t, K, X, xy))
    r, Jc, Jp = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3))
    lib().oracle_reproj_block(*(C.c_void_p(a.ctypes.data) for a in (q, t, K, X, xy, r, Jc, Jp)))
    return r, Jc, Jp


def depth_block(q, t, X, d, shift=0.0, logscale=0.0):
    q, t, X = (np.ascontiguousarray(a, np.float64) for a in (q, t, X))
    r, Jc, Jp = np.zeros(1), np.zeros((1, 6)), np.zeros((1, 3))
    ok = lib().oracle_depth_block(
        q.ctypes.data, t.ctypes.data, X.ctypes.data, float(d), float(shift), float(logscale),
        r.ctypes.data, Jc.ctypes.data, Jp.ctypes.data,
    )
    return int(ok), r, Jc, Jp


def quat_plus(q, d):
    q, d = np.ascontiguousarray(q, np.float64), np.ascontiguousarray(d, np.float64)
    out = np.zeros(4)
    lib().oracle_quat_plus(C.c_void_p(q.ctypes.data), C.c_void_p(d.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def triangulate_tracks(tr: Tracks) -> np.ndarray:
    xyz = np.zeros((tr.n_tracks, 3))
    ct = tr.c_tracks()
    lib().oracle_triangulate_tracks(C.byref(ct), C.c_void_p(xyz.ctypes.data))
    return xyz


def filter_tracks(tr: Tracks, xyz: np.ndarray):
    xyz = np.ascontiguousarray(xyz, np.float64)
    ang = np.zeros(tr.n_tracks)
    err = np.zeros(tr.n_el)
    front = np.zeros(tr.n_el, np.uint8)
    ct = tr.c_tracks()
    lib().oracle_filter_tracks(
        C.byref(ct), C.c_void_p(xyz.ctypes.data), C.c_void_p(ang.ctypes.data), C.c_void_p(err.ctypes.data),
        C.c_void_p(front.ctypes.data),
    )
    return ang, err, front.astype(bool)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def make_allreduce(fn):
    """Wrap a python callable fn(np_array_view) -> None (in-place sum over ranks)."""

    def _cb(user, buf, count, on_device, stream):
        try:
            a = np.ctypeslib.as_array(buf, shape=(count,))
            fn(a)
            return 0
        except Exception:  # pragma: no cover
            return -1

    return ALLREDUCE_FN(_cb)


def available_cpus() -> int:
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))
