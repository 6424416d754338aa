"""track_graph_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent NumPy restatement of the track-graph logic that the reference reaches through
``pycolmap.IncrementalTriangulator`` (reference mpsfm/sfm/mapper/triangulator.py:32-48 options, :88-100
``triangulate_image`` / ``complete_and_merge_*``, :123 ``retriangulate(options, ignore_ids)``; callers
mpsfm/sfm/mapper/base.py:434, 448, 482-485).  The C++ behind those calls lives in the COLMAP fork
``Zador-Pataki/colmap`` at an unpinned HEAD (docker/install_colmap.sh:28) and is NOT in the reference tree:
**parity unpinned**.  What is restated here is the published algorithm of upstream COLMAP 3.11, as recalled:

  src/colmap/estimators/triangulation.cc     TriangulationEstimator::{Estimate, Residuals}, EstimateTriangulation
  src/colmap/optim/loransac.h                LORANSAC<Estimator, LocalEstimator, SupportMeasurer, Sampler>::Estimate
  src/colmap/optim/ransac.h                  RANSAC::ComputeNumTrials
  src/colmap/optim/support_measurement.cc    InlierSupportMeasurer (count, then residual sum)
  src/colmap/optim/combination_sampler.cc    CombinationSampler (lexicographic pairs)
  src/colmap/geometry/triangulation.cc       TriangulatePoint (SVD), TriangulateMultiViewPoint (eigen),
                                             CalculateTriangulationAngle
  src/colmap/scene/projection.cc             CalculateSquaredReprojectionError, CalculateNormalizedAngularError,
                                             HasPointPositiveDepth
  src/colmap/sfm/incremental_triangulator.cc TriangulateImage, CompleteImage, Complete, Merge, Retriangulate,
                                             Find, Create, Continue
  src/colmap/scene/observation_manager.cc    AddPoint3D / AddObservation / DeletePoint3D / MergePoints3D,
                                             num_tri_corrs of the image pairs (kept live)

It shares no code with csrc/tri_math.h / csrc/triangulator.hip: the linear algebra is numpy.linalg (SVD / eigh where
the HIP path runs a Jacobi eigen-solver on the normal matrix), the control flow is written from the upstream sources'
structure (one function per upstream function), and the scene lives in plain Python containers.  Only ``tests/`` may
import this module.

Addressing (same as include/mpsfm_hip.h, row f2): a keypoint is its global index kp_start[image] + point2D_idx; points
are numbered by their position in the state's xyz array and new points continue that range; the operation log lists
(ADD_POINT, id, elements, xyz) / (ADD_OBS, id, keypoint) / (DELETE_POINT, id) in the order they happen.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

EPS = float(np.finfo(np.float64).eps)
ADD_POINT, ADD_OBS, DELETE_POINT = 0, 1, 2
ANGULAR_ERROR, REPROJECTION_ERROR = 0, 1

DEFAULT_OPTIONS = dict(  # IncrementalTriangulator::Options defaults
    max_transitivity=1, create_max_angle_error=2.0, continue_max_angle_error=2.0, merge_max_reproj_error=4.0,
    complete_max_reproj_error=4.0, complete_max_transitivity=5, re_max_angle_error=5.0, re_min_ratio=0.2, re_max_trials=1,
    min_angle=1.5, ignore_two_view_tracks=True)


# ---- geometry -----------------------------------------------------------------------------------------------------------
def quat_to_R(q):
    """Eigen quaternion (x, y, z, w) -> rotation matrix."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def triangulate_point(P1, P2, x1, x2):
    """TriangulatePoint: DLT of two normalised points, null vector by SVD."""
    A = np.stack([x1[0] * P1[2] - P1[0], x1[1] * P1[2] - P1[1], x2[0] * P2[2] - P2[0], x2[1] * P2[2] - P2[1]])
    v = np.linalg.svd(A)[2][3]
    if v[3] == 0:
        return None
    return v[:3] / v[3]


def triangulate_multi_view_point(Ps, xs):
    """TriangulateMultiViewPoint: smallest eigenvector of sum (P - x x^T P)^T (P - x x^T P), x the unit ray."""
    A = np.zeros((4, 4))
    for P, x in zip(Ps, xs):
        r = np.array([x[0], x[1], 1.0])
        r /= np.linalg.norm(r)
        term = P - np.outer(r, r) @ P
        A += term.T @ term
    w, V = np.linalg.eigh(A)
    v = V[:, 0]
    if v[3] == 0:
        return None
    return v[:3] / v[3]


def has_point_positive_depth(P, X):
    return P[2, :3] @ X + P[2, 3] >= EPS


def calculate_triangulation_angle(C1, C2, X):
    b2 = np.sum((C1 - C2) ** 2)
    r1 = np.sum((X - C1) ** 2)
    r2 = np.sum((X - C2) ** 2)
    den = 2.0 * math.sqrt(r1 * r2)
    if den == 0.0:
        return 0.0
    ang = abs(math.acos(min(1.0, max(-1.0, (r1 + r2 - b2) / den))))
    return min(ang, math.pi - ang)


def calculate_normalized_angular_error(xn, X, P):
    r1 = np.array([xn[0], xn[1], 1.0])
    r2 = P[:, :3] @ X + P[:, 3]
    c = (r1 / np.linalg.norm(r1)) @ (r2 / np.linalg.norm(r2))
    return math.acos(min(1.0, max(-1.0, c)))


def calculate_squared_reprojection_error(xy, X, P, K):
    pc = P[:, :3] @ X + P[:, 3]
    if pc[2] < EPS:
        return float(np.finfo(np.float64).max)
    u = K[0] * pc[0] / pc[2] + K[2]
    v = K[1] * pc[1] / pc[2] + K[3]
    return (u - xy[0]) ** 2 + (v - xy[1]) ** 2


# ---- estimator + LORANSAC -------------------------------------------------------------------------------------------------
@dataclass
class View:
    """PointData + PoseData of one observation."""
    xy: np.ndarray          # pixel
    xn: np.ndarray          # camera.CamFromImg(xy)
    P: np.ndarray           # 3x4 cam_from_world
    C: np.ndarray           # projection centre
    K: np.ndarray           # PINHOLE fx fy cx cy


@dataclass
class RansacOptions:
    max_error: float
    min_tri_angle: float = 0.0
    residual_type: int = ANGULAR_ERROR
    confidence: float = 0.9999
    max_num_trials: int = 10000
    min_num_trials: int = 0
    dyn_num_trials_multiplier: float = 3.0


def estimator_estimate(views, min_tri_angle):
    """TriangulationEstimator::Estimate -> list of models (0 or 1)."""
    if len(views) == 2:
        a, b = views
        X = triangulate_point(a.P, b.P, a.xn, b.xn)
        if (X is not None and has_point_positive_depth(a.P, X) and has_point_positive_depth(b.P, X)
                and calculate_triangulation_angle(a.C, b.C, X) >= min_tri_angle):
            return [X]
        return []
    X = triangulate_multi_view_point([v.P for v in views], [v.xn for v in views])
    if X is None:
        return []
    for v in views:
        if not has_point_positive_depth(v.P, X):
            return []
    for i in range(len(views)):
        for j in range(i):
            if calculate_triangulation_angle(views[i].C, views[j].C, X) >= min_tri_angle:
                return [X]
    return []


def estimator_residuals(views, X, residual_type):
    if residual_type == REPROJECTION_ERROR:
        return np.array([calculate_squared_reprojection_error(v.xy, X, v.P, v.K) for v in views])
    return np.array([calculate_normalized_angular_error(v.xn, X, v.P) ** 2 for v in views])


def compute_num_trials(num_inliers, num_samples, confidence, multiplier):
    nom = 1.0 - confidence
    if nom <= 0:
        return 2**63 - 1
    denom = 1.0 - (num_inliers / float(num_samples)) ** 2   # kMinNumSamples = 2
    if denom <= 0:
        return 1
    if denom == 1.0:
        return 2**63 - 1
    return int(math.ceil(math.log(nom) / math.log(denom) * multiplier))


def _support(residuals, max_residual):
    inl = residuals <= max_residual
    return int(inl.sum()), float(residuals[inl].sum())


def _left_better(a, b):
    return a[0] > b[0] or (a[0] == b[0] and a[1] < b[1])


@dataclass
class Report:
    success: bool = False
    num_trials: int = 0
    model: np.ndarray | None = None
    inlier_mask: np.ndarray | None = None
    support: tuple = (0, float(np.finfo(np.float64).max))
    margin: float = math.inf   # diagnostics: smallest relative distance of a final residual to the threshold


def loransac_estimate(views, o: RansacOptions) -> Report:
    """LORANSAC<TriangulationEstimator, TriangulationEstimator, InlierSupportMeasurer, CombinationSampler>::Estimate."""
    n = len(views)
    rep = Report()
    if n < 2:
        return rep
    best_support = rep.support
    best_model = None
    abort = False
    max_residual = o.max_error * o.max_error
    pairs = ((i, j) for i in range(n) for j in range(i + 1, n))   # CombinationSampler: lexicographic
    max_num_trials = min(o.max_num_trials, n * (n - 1) // 2)
    dyn_max_num_trials = max_num_trials
    rep.num_trials = 0
    while rep.num_trials < max_num_trials:
        if abort:
            rep.num_trials += 1
            break
        i, j = next(pairs)
        for sample_model in estimator_estimate([views[i], views[j]], o.min_tri_angle):
            residuals = estimator_residuals(views, sample_model, o.residual_type)
            support = _support(residuals, max_residual)
            if _left_better(support, best_support):
                best_support, best_model = support, sample_model
                if support[0] > 2:
                    for _ in range(10):   # kMaxNumLocalTrials
                        inl = [v for v, r in zip(views, residuals) if r <= max_residual]
                        prev_best = best_support[0]
                        best_local_residuals = None
                        for local_model in estimator_estimate(inl, o.min_tri_angle):
                            residuals = estimator_residuals(views, local_model, o.residual_type)
                            local_support = _support(residuals, max_residual)
                            if _left_better(local_support, best_support):
                                best_support, best_model = local_support, local_model
                                best_local_residuals = residuals
                        if best_support[0] <= prev_best:
                            break
                        residuals = best_local_residuals
                dyn_max_num_trials = compute_num_trials(best_support[0], n, o.confidence, o.dyn_num_trials_multiplier)
            if rep.num_trials >= dyn_max_num_trials and rep.num_trials >= o.min_num_trials:
                abort = True
                break
        rep.num_trials += 1
    rep.support, rep.model = best_support, best_model
    if best_support[0] < 2:
        return rep
    rep.success = True
    residuals = estimator_residuals(views, best_model, o.residual_type)
    rep.inlier_mask = residuals <= max_residual
    with np.errstate(divide="ignore", invalid="ignore"):
        rep.margin = float(np.min(np.abs(residuals - max_residual) / max_residual))
    return rep


# ---- the incremental triangulator --------------------------------------------------------------------------------------
@dataclass
class _Point:
    xyz: np.ndarray
    elements: list = field(default_factory=list)


class TrackGraphOracle:
    def __init__(self, kp_start, kp_xy, intr, corr_start, corr_kp):
        self.kp_start = np.asarray(kp_start, np.int64)
        self.kp_xy = np.asarray(kp_xy, np.float64).reshape(-1, 2)
        self.intr = np.asarray(intr, np.float64).reshape(-1, 4)
        self.corr_start = np.asarray(corr_start, np.int64)
        self.corr_kp = np.asarray(corr_kp, np.int64)
        self.n_images = len(self.kp_start) - 1
        self.kp_image = np.repeat(np.arange(self.n_images), np.diff(self.kp_start))
        self.re_num_trials = {}
        self.merge_trials = {}
        self.ops = []
        self.decision_margin = math.inf
        # total correspondences per image pair (ObservationManager constructor)
        self.pair_total = {}
        for kp in range(len(self.kp_image)):
            for c in self.corrs_of(kp):
                i1, i2 = int(self.kp_image[kp]), int(self.kp_image[c])
                if i1 < i2:
                    self.pair_total[(i1, i2)] = self.pair_total.get((i1, i2), 0) + 1

    def corrs_of(self, kp):
        return [int(c) for c in self.corr_kp[self.corr_start[kp]:self.corr_start[kp + 1]]]

    def set_state(self, registered, quat_xyzw, t, kp_point, xyz):
        self.registered = np.asarray(registered).astype(bool)
        self.P = []
        self.C = []
        for q, tt in zip(np.asarray(quat_xyzw, np.float64).reshape(-1, 4), np.asarray(t, np.float64).reshape(-1, 3)):
            R = quat_to_R(q)
            self.P.append(np.hstack([R, tt[:, None]]))
            self.C.append(-R.T @ tt)
        xyz = np.asarray(xyz, np.float64).reshape(-1, 3)
        self.points = {i: _Point(xyz[i].copy()) for i in range(len(xyz))}
        self.next_id = len(xyz)
        self.kp_point = np.asarray(kp_point, np.int64).copy()
        for kp, p in enumerate(self.kp_point):
            if p >= 0:
                self.points[int(p)].elements.append(kp)
        self.merge_trials = {}
        self.ops = []

    # -- ObservationManager ---------------------------------------------------------------------------------------------
    def has_point(self, kp):
        return self.kp_point[kp] >= 0

    def add_point3D(self, xyz, elements):
        pid = self.next_id
        self.next_id += 1
        self.points[pid] = _Point(np.array(xyz, np.float64), list(elements))
        for kp in elements:
            self.kp_point[kp] = pid
        self.ops.append((ADD_POINT, pid, tuple(int(e) for e in elements), np.array(xyz, np.float64)))
        return pid

    def add_observation(self, pid, kp):
        self.points[pid].elements.append(kp)
        self.kp_point[kp] = pid
        self.ops.append((ADD_OBS, pid, int(kp)))

    def delete_point3D(self, pid):
        for kp in self.points[pid].elements:
            self.kp_point[kp] = -1
        del self.points[pid]
        self.ops.append((DELETE_POINT, pid))

    def merge_points3D(self, id1, id2):
        p1, p2 = self.points[id1], self.points[id2]
        l1, l2 = len(p1.elements), len(p2.elements)
        xyz = (l1 * p1.xyz + l2 * p2.xyz) / (l1 + l2)
        elements = p1.elements + p2.elements
        self.delete_point3D(id1)
        self.delete_point3D(id2)
        return self.add_point3D(xyz, elements)

    def num_tri_corrs(self, i1, i2):
        """image_pair_stats_[pair].num_tri_corrs, which upstream keeps live: correspondences between the two images whose
        two observations belong to the same 3-D point."""
        n = 0
        for kp in range(self.kp_start[i1], self.kp_start[i1 + 1]):
            if self.kp_point[kp] < 0:
                continue
            for c in self.corrs_of(kp):
                if self.kp_image[c] == i2 and self.kp_point[c] == self.kp_point[kp]:
                    n += 1
        return n

    def is_two_view_observation(self, kp):
        c = self.corrs_of(kp)
        return len(c) == 1 and len(self.corrs_of(c[0])) == 1

    def view(self, kp):
        im = int(self.kp_image[kp])
        K, xy = self.intr[im], self.kp_xy[kp]
        return View(xy=xy, xn=np.array([(xy[0] - K[2]) / K[0], (xy[1] - K[3]) / K[1]]), P=self.P[im], C=self.C[im], K=K)

    # -- Find / Create / Continue -------------------------------------------------------------------------------------
    def find(self, kp):
        corrs, num_triangulated = [], 0
        for c in self.corrs_of(kp):
            if not self.registered[self.kp_image[c]]:
                continue
            corrs.append(c)
            if self.has_point(c):
                num_triangulated += 1
        return num_triangulated, corrs

    def _estimate(self, kps, tri_options):
        rep = loransac_estimate([self.view(k) for k in kps], tri_options)
        if rep.success:
            self.decision_margin = min(self.decision_margin, rep.margin)
        return rep

    def create(self, o, corrs_data):
        create_corrs = [kp for kp in corrs_data if not self.has_point(kp)]
        if len(create_corrs) < 2:
            return 0
        if o["ignore_two_view_tracks"] and len(create_corrs) == 2 and self.is_two_view_observation(create_corrs[0]):
            return 0
        tri = RansacOptions(max_error=math.radians(o["create_max_angle_error"]), min_tri_angle=math.radians(o["min_angle"]),
                            residual_type=ANGULAR_ERROR)
        if len(create_corrs) <= 15:   # kExhaustiveSamplingThreshold
            tri.min_num_trials = len(create_corrs) * (len(create_corrs) - 1) // 2
        rep = self._estimate(create_corrs, tri)
        if not rep.success:
            return 0
        track = [kp for kp, m in zip(create_corrs, rep.inlier_mask) if m]
        self.add_point3D(rep.model, track)
        if len(create_corrs) - len(track) >= 3:   # kMinRecursiveTrackLength
            return len(track) + self.create(o, create_corrs)
        return len(track)

    def continue_(self, max_angle_error_deg, ref, corrs_data):
        if self.has_point(ref):
            return 0
        best, best_kp = float(np.finfo(np.float64).max), None
        v = self.view(ref)
        for kp in corrs_data:
            if not self.has_point(kp):
                continue
            err = calculate_normalized_angular_error(v.xn, self.points[int(self.kp_point[kp])].xyz, v.P)
            if err < best:
                best, best_kp = err, kp
        lim = math.radians(max_angle_error_deg)
        if best_kp is not None:
            self.decision_margin = min(self.decision_margin, abs(best - lim) / lim)
        if best_kp is not None and best <= lim:
            self.add_observation(int(self.kp_point[best_kp]), ref)
            return 1
        return 0

    # -- public operations ------------------------------------------------------------------------------------------------
    def _options(self, options):
        o = dict(DEFAULT_OPTIONS)
        o.update(options or {})
        assert o["max_transitivity"] == 1
        return o

    def triangulate_image(self, options, image):
        o = self._options(options)
        self.ops = []
        num_tris = 0
        if not self.registered[image]:
            return 0
        for kp in range(self.kp_start[image], self.kp_start[image + 1]):
            num_triangulated, corrs = self.find(kp)
            if not corrs:
                continue
            if num_triangulated == 0:
                num_tris += self.create(o, corrs + [kp])
            else:
                num_tris += self.continue_(o["continue_max_angle_error"], kp, corrs)
                num_tris += self.create(o, corrs + [kp])
        return num_tris

    def complete_image(self, options, image):
        o = self._options(options)
        self.ops = []
        num_tris = 0
        if not self.registered[image]:
            return 0
        # one options object for the whole loop, as upstream: min_num_trials set for a short track stays for the next
        tri = RansacOptions(max_error=o["complete_max_reproj_error"], min_tri_angle=math.radians(o["min_angle"]),
                            residual_type=REPROJECTION_ERROR)
        for kp in range(self.kp_start[image], self.kp_start[image + 1]):
            if self.has_point(kp):
                num_tris += self.complete(o, int(self.kp_point[kp]))
                continue
            if o["ignore_two_view_tracks"] and self.is_two_view_observation(kp):
                continue
            num_triangulated, corrs = self.find(kp)
            if num_triangulated or not corrs:
                continue
            corrs = corrs + [kp]
            if len(corrs) <= 15:
                tri.min_num_trials = len(corrs) * (len(corrs) - 1) // 2
            rep = self._estimate(corrs, tri)
            if not rep.success:
                continue
            track = [k for k, m in zip(corrs, rep.inlier_mask) if m]
            self.add_point3D(rep.model, track)
            num_tris += len(track)
        return num_tris

    def complete(self, o, pid):
        num_completed = 0
        if pid not in self.points:
            return 0
        max_sq = o["complete_max_reproj_error"] ** 2
        point = self.points[pid]
        queue = list(point.elements)
        for transitivity in range(o["complete_max_transitivity"]):
            if not queue:
                break
            prev_queue, queue = queue, []
            for q in prev_queue:
                for c in self.corrs_of(q):
                    im = int(self.kp_image[c])
                    if not self.registered[im] or self.has_point(c):
                        continue
                    e = calculate_squared_reprojection_error(self.kp_xy[c], point.xyz, self.P[im], self.intr[im])
                    self.decision_margin = min(self.decision_margin, abs(e - max_sq) / max_sq)
                    if e > max_sq:
                        continue
                    self.add_observation(pid, c)
                    if transitivity < o["complete_max_transitivity"] - 1:
                        queue.append(c)
                    num_completed += 1
        return num_completed

    def merge(self, o, pid):
        if pid not in self.points:
            return 0
        max_sq = o["merge_max_reproj_error"] ** 2
        point = self.points[pid]
        for el in list(point.elements):
            for c in self.corrs_of(el):
                if not self.registered[self.kp_image[c]]:
                    continue
                other = int(self.kp_point[c])
                if other < 0 or other == pid or other in self.merge_trials.setdefault(pid, set()):
                    continue
                corr_point = self.points[other]
                self.merge_trials[pid].add(other)
                self.merge_trials.setdefault(other, set()).add(pid)
                l1, l2 = len(point.elements), len(corr_point.elements)
                merged_xyz = (l1 * point.xyz + l2 * corr_point.xyz) / (l1 + l2)
                success = True
                for tr in (point.elements, corr_point.elements):
                    for kp in tr:
                        im = int(self.kp_image[kp])
                        e = calculate_squared_reprojection_error(self.kp_xy[kp], merged_xyz, self.P[im], self.intr[im])
                        self.decision_margin = min(self.decision_margin, abs(e - max_sq) / max_sq)
                        if e > max_sq:
                            success = False
                            break
                    if not success:
                        break
                if success:
                    num_merged = l1 + l2
                    merged = self.merge_points3D(pid, other)
                    rec = self.merge(o, merged)
                    return rec if rec > 0 else num_merged
        return 0

    def complete_tracks(self, options, point_ids=None):
        o = self._options(options)
        self.ops = []
        ids = list(range(self.next_id)) if point_ids is None else [int(p) for p in point_ids]
        return sum(self.complete(o, p) for p in ids)

    def merge_tracks(self, options, point_ids=None):
        o = self._options(options)
        self.ops = []
        ids = list(range(self.next_id)) if point_ids is None else [int(p) for p in point_ids]
        return sum(self.merge(o, p) for p in ids)

    def retriangulate(self, options, ignore_images=()):
        """Retriangulate; upstream walks an unordered map of image pairs — here (as in the HIP engine) ascending pair ids —
        and the fork's `ignore_image_ids` (source absent) is taken as: skip a pair when either image is listed."""
        o = self._options(options)
        self.ops = []
        ignore = set(int(i) for i in ignore_images)
        num_tris = 0
        for (i1, i2) in sorted(self.pair_total):
            tri_ratio = self.num_tri_corrs(i1, i2) / float(self.pair_total[(i1, i2)])
            if tri_ratio >= o["re_min_ratio"]:
                continue
            if i1 in ignore or i2 in ignore:
                continue
            if not self.registered[i1] or not self.registered[i2]:
                continue
            if self.re_num_trials.get((i1, i2), 0) >= o["re_max_trials"]:
                continue
            self.re_num_trials[(i1, i2)] = self.re_num_trials.get((i1, i2), 0) + 1
            for kp1 in range(self.kp_start[i1], self.kp_start[i1 + 1]):   # FindCorrespondencesBetweenImages
                for kp2 in self.corrs_of(kp1):
                    if self.kp_image[kp2] != i2:
                        continue
                    h1, h2 = self.has_point(kp1), self.has_point(kp2)
                    if h1 and h2:
                        continue
                    if h1 and not h2:
                        num_tris += self.continue_(o["re_max_angle_error"], kp2, [kp1])
                    elif not h1 and h2:
                        num_tris += self.continue_(o["re_max_angle_error"], kp1, [kp2])
                    else:
                        num_tris += self.create(o, [kp1, kp2])   # the plain options: no larger threshold for new points
        return num_tris
