"""Flat bundle-adjustment problem descriptor and its ctypes mirror of include/mpsfm_hip.h.

The descriptor is exactly what ``Optimizer.__build_problem`` gathers before it hands the
problem to Ceres (reference mpsfm/sfm/mapper/bundle_adjustment.py:67-185): poses with gauge
flags, points, one reprojection block per observation and one log-depth block per valid prior.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

LOSS_TRIVIAL, LOSS_SOFT_L1, LOSS_CAUCHY = 0, 1, 2
LOSS_BY_NAME = {"trivial": LOSS_TRIVIAL, "softl1": LOSS_SOFT_L1, "soft_l1": LOSS_SOFT_L1, "cauchy": LOSS_CAUCHY}

MAX_TRACE = 64

TERMINATION_NAMES = {
    0: "function_tolerance",
    1: "gradient_tolerance",
    2: "parameter_tolerance",
    3: "max_iterations",
    4: "min_radius",
    5: "invalid_steps",
    6: "no_variables",
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p)


class CProblem(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32),
        ("n_pts", C.c_int32),
        ("n_intr", C.c_int32),
        ("cam_intr", C.c_void_p),
        ("cam_intr_idx", C.c_void_p),
        ("pose_const", C.c_void_p),
        ("gauge_axis_cam", C.c_int32),
        ("pt_const", C.c_void_p),
        ("n_obs", C.c_int64),
        ("obs_cam", C.c_void_p),
        ("obs_pt", C.c_void_p),
        ("obs_xy", C.c_void_p),
        ("reproj_loss_type", C.c_int32),
        ("reproj_loss_scale", C.c_double),
        ("reproj_loss_magnitude", C.c_double),
        ("n_dobs", C.c_int64),
        ("dobs_cam", C.c_void_p),
        ("dobs_pt", C.c_void_p),
        ("dobs_depth", C.c_void_p),
        ("dobs_magnitude", C.c_void_p),
        ("dobs_param", C.c_void_p),
        ("depth_loss_type", C.c_int32),
        ("shift_logscale", C.c_void_p),
    ]


class CState(C.Structure):
    _fields_ = [("cam_quat_xyzw", C.c_void_p), ("cam_t", C.c_void_p), ("pts", C.c_void_p)]


class COptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("verbose", C.c_int32),
        ("allreduce", ALLREDUCE_FN),
        ("allreduce_user", C.c_void_p),
        ("world_size", C.c_int32),
        ("rank", C.c_int32),
        ("use_rccl", C.c_int32),
        ("comm_id", C.c_uint8 * 128),
    ]


class CSummary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("fixed_cost", C.c_double),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("num_residual_blocks", C.c_int64),
        ("num_residual_evals", C.c_int64),
        ("num_jacobian_evals", C.c_int64),
        ("reduced_dim", C.c_int32),
        ("final_radius", C.c_double),
        ("time_total_s", C.c_double),
        ("time_linearize_s", C.c_double),
        ("time_dense_s", C.c_double),
        ("time_update_s", C.c_double),
        ("trace_len", C.c_int32),
        ("trace_cost", C.c_double * MAX_TRACE),
        ("trace_radius", C.c_double * MAX_TRACE),
        ("trace_accepted", C.c_uint8 * MAX_TRACE),
    ]

    def to_dict(self) -> dict:
        n = self.trace_len
        return {
            "initial_cost": self.initial_cost,
            "final_cost": self.final_cost,
            "fixed_cost": self.fixed_cost,
            "num_iterations": self.num_iterations,
            "num_successful_steps": self.num_successful_steps,
            "num_unsuccessful_steps": self.num_unsuccessful_steps,
            "termination": TERMINATION_NAMES.get(self.termination, str(self.termination)),
            "num_residual_blocks": self.num_residual_blocks,
            "num_residual_evals": self.num_residual_evals,
            "num_jacobian_evals": self.num_jacobian_evals,
            "reduced_dim": self.reduced_dim,
            "final_radius": self.final_radius,
            "time_total_s": self.time_total_s,
            "time_linearize_s": self.time_linearize_s,
            "time_dense_s": self.time_dense_s,
            "time_update_s": self.time_update_s,
            "trace_cost": [self.trace_cost[i] for i in range(n)],
            "trace_radius": [self.trace_radius[i] for i in range(n)],
            "trace_accepted": [int(self.trace_accepted[i]) for i in range(n)],
        }


class CTracks(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32),
        ("n_tracks", C.c_int32),
        ("n_intr", C.c_int32),
        ("cam_quat_xyzw", C.c_void_p),
        ("cam_t", C.c_void_p),
        ("cam_intr", C.c_void_p),
        ("cam_intr_idx", C.c_void_p),
        ("track_start", C.c_void_p),
        ("el_cam", C.c_void_p),
        ("el_xy", C.c_void_p),
    ]


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size > 0 else None


@dataclass
class BAProblem:
    """Flat problem + mutable state (NumPy, float64 / int32)."""

    cam_quat: np.ndarray  # [Nc,4] xyzw  (state)
    cam_t: np.ndarray  # [Nc,3]       (state)
    pts: np.ndarray  # [Np,3]       (state)
    cam_intr: np.ndarray  # [Nk,4] fx fy cx cy
    cam_intr_idx: np.ndarray  # [Nc]
    pose_const: np.ndarray  # [Nc] uint8
    pt_const: np.ndarray  # [Np] uint8
    obs_cam: np.ndarray
    obs_pt: np.ndarray
    obs_xy: np.ndarray  # [No,2]
    gauge_axis_cam: int = -1
    reproj_loss_type: int = LOSS_SOFT_L1
    reproj_loss_scale: float = 1.5
    reproj_loss_magnitude: float = 1.0
    dobs_cam: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    dobs_pt: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    dobs_depth: np.ndarray = field(default_factory=lambda: np.zeros(0))
    dobs_magnitude: np.ndarray = field(default_factory=lambda: np.zeros(0))
    dobs_param: np.ndarray = field(default_factory=lambda: np.zeros(0))
    depth_loss_type: int = LOSS_CAUCHY
    shift_logscale: np.ndarray | None = None

    def __post_init__(self):
        self.cam_quat = _arr(self.cam_quat, np.float64, (-1, 4))
        self.cam_t = _arr(self.cam_t, np.float64, (-1, 3))
        self.pts = _arr(self.pts, np.float64, (-1, 3))
        self.cam_intr = _arr(self.cam_intr, np.float64, (-1, 4))
        self.cam_intr_idx = _arr(self.cam_intr_idx, np.int32)
        self.pose_const = _arr(self.pose_const, np.uint8)
        self.pt_const = _arr(self.pt_const, np.uint8)
        self.obs_cam = _arr(self.obs_cam, np.int32)
        self.obs_pt = _arr(self.obs_pt, np.int32)
        self.obs_xy = _arr(self.obs_xy, np.float64, (-1, 2))
        self.dobs_cam = _arr(self.dobs_cam, np.int32)
        self.dobs_pt = _arr(self.dobs_pt, np.int32)
        self.dobs_depth = _arr(self.dobs_depth, np.float64)
        self.dobs_magnitude = _arr(self.dobs_magnitude, np.float64)
        self.dobs_param = _arr(self.dobs_param, np.float64)
        if self.shift_logscale is not None:
            self.shift_logscale = _arr(self.shift_logscale, np.float64, (-1, 2))
        self.validate()

    # -- sizes ------------------------------------------------------------------------
    @property
    def n_cams(self):
        return self.cam_quat.shape[0]

    @property
    def n_pts(self):
        return self.pts.shape[0]

    @property
    def n_obs(self):
        return self.obs_cam.shape[0]

    @property
    def n_dobs(self):
        return self.dobs_cam.shape[0]

    @property
    def n_residual_blocks(self):
        return self.n_obs + self.n_dobs

    def validate(self):
        nc, npt = self.n_cams, self.n_pts
        if self.cam_t.shape[0] != nc or self.cam_intr_idx.shape[0] != nc or self.pose_const.shape[0] != nc:
            raise ValueError("camera arrays disagree on n_cams")
        if self.pt_const.shape[0] != npt:
            raise ValueError("pt_const length != n_pts")
        if self.obs_pt.shape[0] != self.n_obs or self.obs_xy.shape[0] != self.n_obs:
            raise ValueError("observation arrays disagree on n_obs")
        nd = self.n_dobs
        for a in (self.dobs_pt, self.dobs_depth, self.dobs_magnitude, self.dobs_param):
            if a.shape[0] != nd:
                raise ValueError("depth observation arrays disagree on n_dobs")
        if self.n_obs and (self.obs_cam.min() < 0 or self.obs_cam.max() >= nc):
            raise ValueError("obs_cam out of range")
        if self.n_obs and (self.obs_pt.min() < 0 or self.obs_pt.max() >= npt):
            raise ValueError("obs_pt out of range")
        if nd and (self.dobs_cam.min() < 0 or self.dobs_cam.max() >= nc):
            raise ValueError("dobs_cam out of range")
        if nd and (self.dobs_pt.min() < 0 or self.dobs_pt.max() >= npt):
            raise ValueError("dobs_pt out of range")
        if nc and (self.cam_intr_idx.min() < 0 or self.cam_intr_idx.max() >= self.cam_intr.shape[0]):
            raise ValueError("cam_intr_idx out of range")
        if self.shift_logscale is not None and self.shift_logscale.shape[0] != nc:
            raise ValueError("shift_logscale must be [n_cams,2]")
        if not (-1 <= self.gauge_axis_cam < max(nc, 1)):
            raise ValueError("gauge_axis_cam out of range")

    def copy(self) -> "BAProblem":
        import copy

        return copy.deepcopy(self)

    # -- ctypes views (arrays stay owned by self) ---------------------------------------
    def c_problem(self) -> CProblem:
        p = CProblem()
        p.n_cams, p.n_pts, p.n_intr = self.n_cams, self.n_pts, self.cam_intr.shape[0]
        p.cam_intr = _ptr(self.cam_intr)
        p.cam_intr_idx = _ptr(self.cam_intr_idx)
        p.pose_const = _ptr(self.pose_const)
        p.gauge_axis_cam = int(self.gauge_axis_cam)
        p.pt_const = _ptr(self.pt_const)
        p.n_obs = self.n_obs
        p.obs_cam, p.obs_pt, p.obs_xy = _ptr(self.obs_cam), _ptr(self.obs_pt), _ptr(self.obs_xy)
        p.reproj_loss_type = int(self.reproj_loss_type)
        p.reproj_loss_scale = float(self.reproj_loss_scale)
        p.reproj_loss_magnitude = float(self.reproj_loss_magnitude)
        p.n_dobs = self.n_dobs
        p.dobs_cam, p.dobs_pt = _ptr(self.dobs_cam), _ptr(self.dobs_pt)
        p.dobs_depth, p.dobs_magnitude, p.dobs_param = (
            _ptr(self.dobs_depth),
            _ptr(self.dobs_magnitude),
            _ptr(self.dobs_param),
        )
        p.depth_loss_type = int(self.depth_loss_type)
        p.shift_logscale = _ptr(self.shift_logscale) if self.shift_logscale is not None else None
        return p

    def c_state(self) -> CState:
        s = CState()
        s.cam_quat_xyzw, s.cam_t, s.pts = _ptr(self.cam_quat), _ptr(self.cam_t), _ptr(self.pts)
        return s


@dataclass
class Tracks:
    """Tracks for the batch triangulation numerics (CSR by track)."""

    cam_quat: np.ndarray
    cam_t: np.ndarray
    cam_intr: np.ndarray
    cam_intr_idx: np.ndarray
    track_start: np.ndarray  # [Nt+1] int64
    el_cam: np.ndarray  # [Ne] int32
    el_xy: np.ndarray  # [Ne,2]

    def __post_init__(self):
        self.cam_quat = _arr(self.cam_quat, np.float64, (-1, 4))
        self.cam_t = _arr(self.cam_t, np.float64, (-1, 3))
        self.cam_intr = _arr(self.cam_intr, np.float64, (-1, 4))
        self.cam_intr_idx = _arr(self.cam_intr_idx, np.int32)
        self.track_start = _arr(self.track_start, np.int64)
        self.el_cam = _arr(self.el_cam, np.int32)
        self.el_xy = _arr(self.el_xy, np.float64, (-1, 2))
        ne = self.el_cam.shape[0]
        if self.track_start.shape[0] < 1 or self.track_start[0] != 0 or self.track_start[-1] != ne:
            raise ValueError("track_start must be CSR offsets covering the element arrays")
        if np.any(np.diff(self.track_start) < 0):
            raise ValueError("track_start must be non-decreasing")
        if ne and (self.el_cam.min() < 0 or self.el_cam.max() >= self.cam_quat.shape[0]):
            raise ValueError("el_cam out of range")

    @property
    def n_tracks(self):
        return self.track_start.shape[0] - 1

    @property
    def n_el(self):
        return self.el_cam.shape[0]

    def c_tracks(self) -> CTracks:
        t = CTracks()
        t.n_cams, t.n_tracks, t.n_intr = self.cam_quat.shape[0], self.n_tracks, self.cam_intr.shape[0]
        t.cam_quat_xyzw, t.cam_t = _ptr(self.cam_quat), _ptr(self.cam_t)
        t.cam_intr, t.cam_intr_idx = _ptr(self.cam_intr), _ptr(self.cam_intr_idx)
        t.track_start, t.el_cam, t.el_xy = _ptr(self.track_start), _ptr(self.el_cam), _ptr(self.el_xy)
        return t


INT_MAX_IRLS = 16


class CIntProblem(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32),
        ("depth_prior", C.c_void_p), ("depth_uncertainty", C.c_void_p), ("valid", C.c_void_p), ("normals", C.c_void_p),
        ("normals_var", C.c_void_p), ("depth_init", C.c_void_p), ("K", C.c_double * 4),
        ("n_sparse", C.c_int32), ("sparse_x", C.c_void_p), ("sparse_y", C.c_void_p), ("sparse_depth3d", C.c_void_p),
        ("sparse_zvar", C.c_void_p),
        ("large_number", C.c_double), ("tol", C.c_double), ("step_size", C.c_double), ("cg_tol", C.c_double),
        ("lambda1", C.c_double), ("lambda2", C.c_double), ("k", C.c_double), ("depth_magnitude_multiplier", C.c_double),
        ("normals_magnitude_multiplier", C.c_double), ("scale_filter_factor", C.c_double),
        ("max_iter", C.c_int32), ("cg_max_iter", C.c_int32), ("scale_filter", C.c_int32),
        ("init", C.c_int32), ("integrated", C.c_int32), ("energy_old", C.c_double), ("wu", C.c_void_p), ("wv", C.c_void_p),
    ]


class CIntSummary(C.Structure):
    _fields_ = [
        ("changed", C.c_int32), ("irls_iterations", C.c_int32), ("cg_iterations_total", C.c_int32), ("integrated_out", C.c_int32),
        ("energy_initial", C.c_double), ("energy_final", C.c_double), ("energy_old_out", C.c_double),
        ("cg_iters", C.c_int32 * INT_MAX_IRLS), ("energies", C.c_double * (INT_MAX_IRLS + 1)), ("ms", C.c_float),
    ]
