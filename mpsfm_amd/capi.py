"""ctypes binding of libmpsfm_hip.so (include/mpsfm_hip.h).

This is the only door between the Python host layer and the HIP kernels.  There is no CPU
fallback: every call fails loudly when the library is missing or no gfx950 device is visible.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .problem import ALLREDUCE_FN, BAProblem, COptions, CProblem, CState, CSummary, CTracks, Tracks

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpsfm_hip.so")

EXPORTS = [
    "mpsfm_abi_version", "mpsfm_last_error", "mpsfm_device_count", "mpsfm_ba_default_options",
    "mpsfm_ba_solve", "mpsfm_ba_create", "mpsfm_ba_set_state", "mpsfm_ba_reset_state",
    "mpsfm_ba_solve_resident", "mpsfm_ba_get_state", "mpsfm_ba_destroy", "mpsfm_ba_eval_cost",
    "mpsfm_ba_sweep_once", "mpsfm_ba_get_reduced_system", "mpsfm_ba_reduced_dim",
    "mpsfm_ba_get_dense_solution", "mpsfm_ba_dense_solve_once", "mpsfm_point_covs",
    "mpsfm_triangulate_tracks", "mpsfm_filter_tracks",
]

_lib = None


class MpsfmHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmpsfm_hip error {code}: {msg}")
        self.code = code


def lib():
    """Loads libmpsfm_hip.so (building it is __graft_entry__.build()'s job, not ours)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpsfmHipError(-2, f"{LIB_PATH} is missing: run `python -m mpsfm_amd.build` (hipcc, gfx950)")
        try:
            # torch wheels bundle their own libamdhip64; if ours (linked against /opt/rocm) is loaded
            # first the process ends up with two HIP runtimes and torch.cuda reports no devices.
            # Importing torch first makes its runtime the process-wide one (same soname).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.mpsfm_last_error.restype = C.c_char_p
        L.mpsfm_ba_destroy.restype = None
        L.mpsfm_ba_default_options.restype = None
        L.mpsfm_ba_sweep_once.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_float)]
        L.mpsfm_ba_dense_solve_once.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.mpsfm_ba_get_reduced_system.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.mpsfm_ba_get_dense_solution.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.mpsfm_ba_eval_cost.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for name in ("mpsfm_ba_set_state", "mpsfm_ba_get_state", "mpsfm_ba_solve_resident"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.mpsfm_ba_reset_state.argtypes = [C.c_void_p]
        L.mpsfm_ba_reduced_dim.argtypes = [C.c_void_p]
        L.mpsfm_ba_destroy.argtypes = [C.c_void_p]
        L.mpsfm_ba_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mpsfm_ba_solve.argtypes = [C.c_void_p] * 4
        L.mpsfm_point_covs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.mpsfm_triangulate_tracks.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.mpsfm_filter_tracks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise MpsfmHipError(rc, (lib().mpsfm_last_error() or b"").decode())


def device_count() -> int:
    return int(lib().mpsfm_device_count())


def default_options(**kw) -> COptions:
    o = COptions()
    lib().mpsfm_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def make_allreduce(fn):
    """fn(ptr:int, count:int, on_device:bool, stream:int) -> None must sum the buffer over ranks."""

    def _cb(user, buf, count, on_device, stream):
        try:
            fn(C.addressof(buf.contents), int(count), bool(on_device), int(stream or 0))
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through C
            import sys

            print(f"[mpsfm_amd] all-reduce hook raised: {e!r}", file=sys.stderr)
            return -1

    return ALLREDUCE_FN(_cb)


class BAHandle:
    """Resident problem: the observation lists, poses and points live in HBM between calls."""

    def __init__(self, prob: BAProblem, options: COptions | None = None):
        self._h = C.c_void_p(None)
        self.prob = prob
        self.options = options if options is not None else default_options()
        cp, cs = prob.c_problem(), prob.c_state()
        _check(lib().mpsfm_ba_create(C.byref(cp), C.byref(cs), C.byref(self.options), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().mpsfm_ba_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_state(self, prob: BAProblem | None = None):
        cs = (prob or self.prob).c_state()
        _check(lib().mpsfm_ba_set_state(self._h, C.byref(cs)))

    def reset_state(self):
        _check(lib().mpsfm_ba_reset_state(self._h))

    def solve(self) -> dict:
        sm = CSummary()
        _check(lib().mpsfm_ba_solve_resident(self._h, C.byref(sm)))
        return sm.to_dict()

    def get_state(self, prob: BAProblem | None = None):
        """Writes the resident poses/points into prob (default: the problem given at creation)."""
        cs = (prob or self.prob).c_state()
        _check(lib().mpsfm_ba_get_state(self._h, C.byref(cs)))

    def eval_cost(self) -> tuple[float, float]:
        a, b = C.c_double(0), C.c_double(0)
        _check(lib().mpsfm_ba_eval_cost(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def reduced_dim(self) -> int:
        return int(lib().mpsfm_ba_reduced_dim(self._h))

    def sweep_once(self, radius: float = 1e4) -> float:
        ms = C.c_float(0)
        _check(lib().mpsfm_ba_sweep_once(self._h, radius, C.byref(ms)))
        return ms.value

    def dense_solve_once(self) -> float:
        ms = C.c_float(0)
        _check(lib().mpsfm_ba_dense_solve_once(self._h, C.byref(ms)))
        return ms.value

    def reduced_system(self):
        n = self.reduced_dim
        S, rhs = np.zeros((n, n)), np.zeros(n)
        _check(lib().mpsfm_ba_get_reduced_system(self._h, S.ctypes.data, rhs.ctypes.data, n))
        return S, rhs

    def dense_solution(self):
        n = self.reduced_dim
        y = np.zeros(n)
        _check(lib().mpsfm_ba_get_dense_solution(self._h, y.ctypes.data, n))
        return y


def ba_solve(prob: BAProblem, options: COptions | None = None) -> dict:
    """One-shot mpsfm_ba_solve: refines prob.cam_quat / cam_t / pts in place."""
    o = options if options is not None else default_options()
    cp, cs, sm = prob.c_problem(), prob.c_state(), CSummary()
    _check(lib().mpsfm_ba_solve(C.byref(cp), C.byref(cs), C.byref(o), C.byref(sm)))
    return sm.to_dict()


def point_covs(prob: BAProblem, device: int = 0) -> np.ndarray:
    covs = np.zeros((prob.n_pts, 3, 3))
    cp, cs = prob.c_problem(), prob.c_state()
    _check(lib().mpsfm_point_covs(C.byref(cp), C.byref(cs), device, covs.ctypes.data))
    return covs


def triangulate_tracks(tr: Tracks, device: int = 0) -> np.ndarray:
    xyz = np.zeros((tr.n_tracks, 3))
    ct = tr.c_tracks()
    _check(lib().mpsfm_triangulate_tracks(C.byref(ct), device, xyz.ctypes.data))
    return xyz


def filter_tracks(tr: Tracks, xyz: np.ndarray, device: int = 0):
    xyz = np.ascontiguousarray(xyz, np.float64)
    ang, err, front = np.zeros(tr.n_tracks), np.zeros(tr.n_el), np.zeros(tr.n_el, np.uint8)
    ct = tr.c_tracks()
    _check(lib().mpsfm_filter_tracks(C.byref(ct), xyz.ctypes.data, device, ang.ctypes.data, err.ctypes.data, front.ctypes.data))
    return ang, err, front.astype(bool)
