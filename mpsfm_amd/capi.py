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
    "mpsfm_ba_get_dense_solution", "mpsfm_ba_dense_solve_once", "mpsfm_ba_dense_plan", "mpsfm_point_covs",
    "mpsfm_triangulate_tracks", "mpsfm_filter_tracks", "mpsfm_integrate_depth", "mpsfm_integrate_depth_batch",
    "mpsfm_integration_variances", "mpsfm_depth_blocks", "mpsfm_comm_unique_id",
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
        L.mpsfm_ba_dense_plan.argtypes = [C.c_void_p, C.c_void_p]
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


def comm_unique_id() -> bytes:
    """128 bytes for COptions.comm_id (ncclGetUniqueId of the RCCL library in the process)."""
    buf = (C.c_uint8 * 128)()
    L = lib()
    L.mpsfm_comm_unique_id.argtypes = [C.c_void_p]
    _check(L.mpsfm_comm_unique_id(C.addressof(buf)))
    return bytes(buf)


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


# Handles still open when the interpreter exits are closed from a Python atexit hook, i.e. BEFORE the C runtime's own exit
# handlers: a handle destroyed later (garbage collection at shutdown, static destruction) calls into a HIP runtime that may already
# be gone — under rocprofv3, whose tool finalisation runs first, that was a segmentation fault at exit.
_open_handles: "weakref.WeakSet" = None


def _track(handle):
    global _open_handles
    import atexit
    import weakref

    if _open_handles is None:
        _open_handles = weakref.WeakSet()

        def _close_all():
            for h in list(_open_handles):
                try:
                    h.close()
                except Exception:  # noqa: BLE001
                    pass

        atexit.register(_close_all)
    _open_handles.add(handle)


class BAHandle:
    """Resident problem: the observation lists, poses and points live in HBM between calls."""

    def __init__(self, prob: BAProblem, options: COptions | None = None):
        self._h = C.c_void_p(None)
        self.prob = prob
        self.options = options if options is not None else default_options()
        cp, cs = prob.c_problem(), prob.c_state()
        _check(lib().mpsfm_ba_create(C.byref(cp), C.byref(cs), C.byref(self.options), C.byref(self._h)))
        _track(self)

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

    def dense_plan(self) -> dict:
        """How the handle factors the reduced camera system (slot order, elimination-tree levels, launch counts)."""
        v = (C.c_int64 * 10)()
        _check(lib().mpsfm_ba_dense_plan(self._h, v))
        keys = ["slots", "tile_columns", "levels", "nd_depth", "inverse_accumulators", "work_items", "tile_products", "inverse_roles",
                "s_blocks", "backsub_launches"]
        return dict(zip(keys, (int(x) for x in v)))

    def sweep_once(self, radius: float = 1e4) -> float:
        ms = C.c_float(0)
        _check(lib().mpsfm_ba_sweep_once(self._h, radius, C.byref(ms)))
        return ms.value

    def sweep_parts(self) -> dict:
        """HIP-event times of the parts of the last sweep_once and the chunk counts behind them."""
        ms, info = (C.c_float * 3)(), (C.c_int64 * 4)()
        lib().mpsfm_ba_sweep_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib().mpsfm_ba_sweep_parts(self._h, ms, info))
        return dict(dense_ms=ms[0], reduce_ms=ms[1], general_ms=ms[2], dense_chunks=int(info[0]), general_chunks=int(info[1]),
                    long_tracks=int(info[2]), reduce_parts=int(info[3]))

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


class CDepthGather(C.Structure):
    _fields_ = [
        ("n_images", C.c_int32), ("map_h", C.c_void_p), ("map_w", C.c_void_p), ("depth_map", C.c_void_p), ("valid_map", C.c_void_p),
        ("sx", C.c_void_p), ("sy", C.c_void_p), ("cam_quat_xyzw", C.c_void_p), ("cam_t", C.c_void_p),
        ("n_obs", C.c_int64), ("obs_img", C.c_void_p), ("obs_xy", C.c_void_p), ("obs_var", C.c_void_p), ("obs_pt", C.c_void_p),
        ("n_pts", C.c_int32), ("pts", C.c_void_p),
        ("scale_filter", C.c_int32), ("scale_filter_factor", C.c_double), ("gross_outliers", C.c_int32), ("multiplier", C.c_double),
    ]


def depth_blocks(depth_maps, valid_maps, sx, sy, cam_quat, cam_t, obs_img, obs_xy, obs_var, obs_pt, pts, scale_filter_factor=1.5,
                 multiplier=2.0, device=0):
    """mpsfm_depth_blocks: the depth-block selection of a whole bundle in one launch.  `depth_maps` / `valid_maps`: one
    [H,W] array per image.  Returns dict(flags uint8 [n], depth, depth3d, magnitude, param, whitened float64 [n])."""
    n_img = len(depth_maps)
    dm = [np.ascontiguousarray(m, np.float64) for m in depth_maps]
    vm = [np.ascontiguousarray(m, np.uint8) for m in valid_maps]
    hh = np.array([m.shape[0] for m in dm], np.int32)
    ww = np.array([m.shape[1] for m in dm], np.int32)
    f64 = lambda a, shape=None: np.ascontiguousarray(a, np.float64).reshape(shape) if shape else np.ascontiguousarray(a, np.float64)  # noqa: E731
    sx, sy, cam_quat, cam_t = f64(sx), f64(sy), f64(cam_quat, (-1, 4)), f64(cam_t, (-1, 3))
    obs_img, obs_pt = np.ascontiguousarray(obs_img, np.int32), np.ascontiguousarray(obs_pt, np.int32)
    obs_xy, obs_var, pts = f64(obs_xy, (-1, 2)), f64(obs_var), f64(pts, (-1, 3))
    n = len(obs_img)
    G = CDepthGather()
    G.n_images = n_img
    G.map_h, G.map_w = hh.ctypes.data, ww.ctypes.data
    dptr = (C.c_void_p * max(n_img, 1))(*[m.ctypes.data for m in dm])
    vptr = (C.c_void_p * max(n_img, 1))(*[m.ctypes.data for m in vm])
    G.depth_map, G.valid_map = C.addressof(dptr), C.addressof(vptr)
    G.sx, G.sy, G.cam_quat_xyzw, G.cam_t = sx.ctypes.data, sy.ctypes.data, cam_quat.ctypes.data, cam_t.ctypes.data
    G.n_obs = n
    G.obs_img, G.obs_xy, G.obs_var, G.obs_pt = obs_img.ctypes.data, obs_xy.ctypes.data, obs_var.ctypes.data, obs_pt.ctypes.data
    G.n_pts, G.pts = len(pts), pts.ctypes.data
    G.scale_filter, G.scale_filter_factor, G.gross_outliers, G.multiplier = 1, float(scale_filter_factor), 0, float(multiplier)
    out = {k: np.zeros(n) for k in ("depth", "depth3d", "magnitude", "param", "whitened")}
    out["flags"] = np.zeros(n, np.uint8)
    L = lib()
    L.mpsfm_depth_blocks.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
    _check(L.mpsfm_depth_blocks(C.byref(G), device, out["flags"].ctypes.data, out["depth"].ctypes.data, out["depth3d"].ctypes.data,
                                out["magnitude"].ctypes.data, out["param"].ctypes.data, out["whitened"].ctypes.data))
    return out


INT_DEFAULT_CONF = dict(
    large_number=1e6, max_iter=10, tol=5e-2, step_size=1, cg_max_iter=5000, cg_tol=1e-3, lambda1=1, lambda2=1, k=1,
    depth_magnitude_multiplier=1, normals_magnitude_multiplier=1, scale_filter=True, scale_filter_factor=1.5,
)


def _int_problem(depth_prior, depth_uncertainty, valid, normals, normals_var, depth_init, K, kps, depth3d, zvars3d, conf,
                 init=True, integrated=False, energy_old=0.0, wu=None, wv=None):
    """Fills a CIntProblem; returns (P, keepalive tuple, (H, W), wu, wv)."""
    from .problem import CIntProblem

    c = dict(INT_DEFAULT_CONF)
    c.update(conf or {})
    f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
    depth_prior, depth_uncertainty, depth_init = f64(depth_prior), f64(depth_uncertainty), f64(depth_init)
    H, W = depth_prior.shape
    valid = np.ascontiguousarray(valid, np.uint8)
    normals, normals_var = f64(normals).reshape(H, W, 3), f64(normals_var).reshape(H, W, 3)
    kps = np.ascontiguousarray(kps, np.int64).reshape(-1, 2)
    sx, sy = np.ascontiguousarray(kps[:, 0], np.int32), np.ascontiguousarray(kps[:, 1], np.int32)
    depth3d, zvars3d = f64(depth3d), f64(zvars3d)
    wu = np.zeros(H * W) if wu is None else f64(wu).copy()
    wv = np.zeros(H * W) if wv is None else f64(wv).copy()
    P = CIntProblem()
    P.H, P.W = H, W
    P.depth_prior, P.depth_uncertainty, P.valid = depth_prior.ctypes.data, depth_uncertainty.ctypes.data, valid.ctypes.data
    P.normals, P.normals_var, P.depth_init = normals.ctypes.data, normals_var.ctypes.data, depth_init.ctypes.data
    P.K = (C.c_double * 4)(*[float(v) for v in K])
    P.n_sparse = len(sx)
    P.sparse_x, P.sparse_y = (sx.ctypes.data, sy.ctypes.data) if len(sx) else (None, None)
    P.sparse_depth3d, P.sparse_zvar = (depth3d.ctypes.data, zvars3d.ctypes.data) if len(sx) else (None, None)
    for k in ("large_number", "tol", "step_size", "cg_tol", "lambda1", "lambda2", "k", "depth_magnitude_multiplier",
              "normals_magnitude_multiplier", "scale_filter_factor"):
        setattr(P, k, float(c[k]))
    P.max_iter, P.cg_max_iter, P.scale_filter = int(c["max_iter"]), int(c["cg_max_iter"]), int(bool(c["scale_filter"]))
    P.init, P.integrated, P.energy_old = int(bool(init)), int(bool(integrated)), float(energy_old or 0.0)
    P.wu, P.wv = wu.ctypes.data, wv.ctypes.data
    keep = (depth_prior, depth_uncertainty, depth_init, valid, normals, normals_var, sx, sy, depth3d, zvars3d)
    return P, keep, (H, W), wu, wv


def integrate_depth(depth_prior, depth_uncertainty, valid, normals, normals_var, depth_init, K, kps, depth3d, zvars3d,
                    conf=None, init=True, integrated=False, energy_old=0.0, wu=None, wv=None, device=0):
    """mpsfm_integrate_depth.  Returns (depth map or None, summary dict, wu, wv)."""
    from .problem import CIntSummary

    P, _keep, (H, W), wu, wv = _int_problem(depth_prior, depth_uncertainty, valid, normals, normals_var, depth_init, K, kps,
                                            depth3d, zvars3d, conf, init, integrated, energy_old, wu, wv)
    out = np.zeros((H, W))
    S = CIntSummary()
    L = lib()
    L.mpsfm_integrate_depth.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    _check(L.mpsfm_integrate_depth(C.byref(P), device, out.ctypes.data, C.byref(S)))
    n = S.irls_iterations
    summary = dict(changed=bool(S.changed), irls_iterations=n, cg_iterations_total=S.cg_iterations_total,
                   integrated=bool(S.integrated_out), energy_old=S.energy_old_out, energy_initial=S.energy_initial,
                   energy_final=S.energy_final, cg_iters=[S.cg_iters[i] for i in range(n)],
                   energies=[S.energies[i] for i in range(n + 1)], ms=S.ms)
    return (out if S.changed else None), summary, wu, wv


def _int_summary_dict(S):
    n = S.irls_iterations
    return dict(changed=bool(S.changed), irls_iterations=n, cg_iterations_total=S.cg_iterations_total,
                integrated=bool(S.integrated_out), energy_old=S.energy_old_out, energy_initial=S.energy_initial,
                energy_final=S.energy_final, cg_iters=[S.cg_iters[i] for i in range(n)],
                energies=[S.energies[i] for i in range(n + 1)], ms=S.ms)


def integrate_depth_batch(items, conf=None, device=0):
    """mpsfm_integrate_depth_batch.  `items`: list of dicts with the arguments of `integrate_depth`
    (depth_prior, depth_uncertainty, valid, normals, normals_var, depth_init, K, kps, depth3d, zvars3d and
    optionally init, integrated, energy_old, wu, wv); all maps of one size.  Returns a list of
    (depth map or None, summary dict, wu, wv) in the same order."""
    from .problem import CIntProblem, CIntSummary

    n = len(items)
    if n == 0:
        return []
    Ps = (CIntProblem * n)()
    Ss = (CIntSummary * n)()
    keep, outs, wus, wvs = [], [], [], []
    for i, it in enumerate(items):
        P, k, (H, W), wu, wv = _int_problem(it["depth_prior"], it["depth_uncertainty"], it["valid"], it["normals"], it["normals_var"],
                                            it["depth_init"], it["K"], it["kps"], it["depth3d"], it["zvars3d"], conf,
                                            it.get("init", True), it.get("integrated", False), it.get("energy_old", 0.0),
                                            it.get("wu"), it.get("wv"))
        Ps[i] = P
        keep.append(k)
        outs.append(np.zeros((H, W)))
        wus.append(wu)
        wvs.append(wv)
    ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    L = lib()
    L.mpsfm_integrate_depth_batch.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    _check(L.mpsfm_integrate_depth_batch(n, C.byref(Ps), device, C.byref(ptrs), C.byref(Ss)))
    return [((outs[i] if Ss[i].changed else None), _int_summary_dict(Ss[i]), wus[i], wvs[i]) for i in range(n)]


def integration_variances(depth_prior, depth_uncertainty, valid, normals, normals_var, depth_checkpoint, K, query_xy,
                          kps=None, depth3d=None, zvars3d=None, use_sparse=False, conf=None, rtol=1e-10, max_iter=50000,
                          device=0, return_field=False):
    """mpsfm_integration_variances: var(log depth) propagated through the integration at integer pixels
    `query_xy` [n,2] (x, y).  Returns (variances [n], summary dict[, field [H,W]])."""
    from .problem import CIntSummary

    empty = np.zeros((0, 2), np.int64)
    P, _keep, (H, W), _, _ = _int_problem(depth_prior, depth_uncertainty, valid, normals, normals_var, depth_checkpoint, K,
                                          empty if kps is None else kps, [] if depth3d is None else depth3d,
                                          [] if zvars3d is None else zvars3d, conf, init=False)
    q = np.ascontiguousarray(query_xy, np.int64).reshape(-1, 2)
    qx, qy = np.ascontiguousarray(q[:, 0], np.int32), np.ascontiguousarray(q[:, 1], np.int32)
    out = np.zeros(len(q))
    field = np.zeros((H, W)) if return_field else None
    S = CIntSummary()
    L = lib()
    L.mpsfm_integration_variances.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.mpsfm_integration_variances(C.byref(P), device, int(bool(use_sparse)), len(q), qx.ctypes.data if len(q) else None,
                                         qy.ctypes.data if len(q) else None, float(rtol), int(max_iter),
                                         out.ctypes.data if len(q) else None, field.ctypes.data if return_field else None,
                                         C.byref(S)))
    summary = dict(converged=bool(S.changed), cg_iterations=S.cg_iterations_total, ms=S.ms)
    return (out, summary, field) if return_field else (out, summary)


class CTriCandidates(C.Structure):
    _fields_ = [("n_candidates", C.c_int64), ("cand_start", C.c_void_p), ("view_cam_from_world", C.c_void_p), ("view_intr", C.c_void_p),
                ("view_xy", C.c_void_p), ("min_tri_angle", C.c_double), ("max_error", C.c_double), ("residual_type", C.c_int32),
                ("min_num_trials", C.c_void_p)]


def tri_estimate_batch(cand_start, view_cam_from_world, view_intr, view_xy, min_tri_angle, max_error, residual_type=0,
                       min_num_trials=None, device=0):
    """mpsfm_tri_estimate_batch: COLMAP's EstimateTriangulation for many candidate tracks in one launch.
    Returns (xyz [n,3], ok [n] bool, inlier [n_views] bool)."""
    cs = np.ascontiguousarray(cand_start, np.int64)
    n = len(cs) - 1
    P = np.ascontiguousarray(view_cam_from_world, np.float64).reshape(-1, 12)
    K = np.ascontiguousarray(view_intr, np.float64).reshape(-1, 4)
    xy = np.ascontiguousarray(view_xy, np.float64).reshape(-1, 2)
    if not (len(P) == len(K) == len(xy) == (int(cs[-1]) if n >= 0 and len(cs) else 0)):
        raise ValueError("view arrays do not match cand_start")
    mt = None if min_num_trials is None else np.ascontiguousarray(min_num_trials, np.int64)
    c = CTriCandidates(n, cs.ctypes.data, P.ctypes.data, K.ctypes.data, xy.ctypes.data, float(min_tri_angle), float(max_error),
                       int(residual_type), None if mt is None else mt.ctypes.data)
    xyz, ok, inl = np.zeros((max(n, 0), 3)), np.zeros(max(n, 0), np.uint8), np.zeros(len(P), np.uint8)
    L = lib()
    L.mpsfm_tri_estimate_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.mpsfm_tri_estimate_batch(C.byref(c), device, xyz.ctypes.data, ok.ctypes.data, inl.ctypes.data))
    return xyz, ok.astype(bool), inl.astype(bool)
