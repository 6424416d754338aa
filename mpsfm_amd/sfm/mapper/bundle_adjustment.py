"""Drop-in ``Optimizer`` over libmpsfm_hip.

Host-side mirror of reference ``mpsfm/sfm/mapper/bundle_adjustment.py`` (class Optimizer, :18-333):
same constructor, configuration keys, method names, argument meaning and return shapes, so that
``MpsfmMapper`` (reference mapper/base.py:171-175, 420-440, 516-617) can use it unchanged.  Where the
reference builds a Ceres problem through pycolmap and calls ``pyceres.solve`` (:85-104, :163-176,
:285-293), this class gathers the same residual blocks into the flat descriptor of
include/mpsfm_hip.h and runs the HIP solver; poses and points are written back in place like
Ceres does through the pybind11 views (:113-122).

There is no CPU fallback here: the default backend is the C ABI; tests may inject another
backend object (``solve(BAProblem) -> dict``, ``point_covs(BAProblem) -> [N,3,3]``).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ...baseclass import BaseClass
from ...problem import LOSS_BY_NAME, LOSS_CAUCHY, LOSS_SOFT_L1, LOSS_TRIVIAL, BAProblem
from ..scene.priorutils import fit_robust_gaussian_mad

_COLMAP_LOSS = {"TRIVIAL": LOSS_TRIVIAL, "SOFT_L1": LOSS_SOFT_L1, "CAUCHY": LOSS_CAUCHY}


def pinhole_params(camera) -> np.ndarray:
    """[fx, fy, cx, cy] of a camera.  The kernels implement COLMAP's PINHOLE projection only (the reference's own
    loader creates nothing else, data_proc/simple.py:53-60); SIMPLE_PINHOLE is widened to (f, f, cx, cy), any other
    model is refused instead of being optimised with a wrong projection."""
    model = getattr(camera, "model", None)
    name = getattr(model, "name", model)
    name = "PINHOLE" if name is None else str(name).split(".")[-1]
    params = np.asarray(camera.params, np.float64)
    if name == "PINHOLE":
        return params[:4]
    if name == "SIMPLE_PINHOLE":
        return np.array([params[0], params[0], params[1], params[2]])
    raise NotImplementedError(f"camera model {name}: libmpsfm_hip implements the PINHOLE reprojection functor only")


class HipBackend:
    """The product backend: include/mpsfm_hip.h through ctypes."""

    def __init__(self, device: int = 0):
        self.device = device

    def solve(self, prob: BAProblem) -> dict:
        from ... import capi

        return capi.ba_solve(prob, capi.default_options(device=self.device))

    def point_covs(self, prob: BAProblem) -> np.ndarray:
        from ... import capi

        return capi.point_covs(prob, self.device)


@dataclass
class FlatProblem:
    """What __build_problem hands to the solver, plus the bookkeeping to write results back."""

    prob: BAProblem
    image_ids: list          # camera index -> image id (bundle images first, in optim_ids order)
    point_ids: list          # point index -> point3D id
    shift_scale: dict        # {imid: [shift, log-scale]} scratch, all zeros (reference :83)
    summary: dict | None = None


class Optimizer(BaseClass):
    """Optimizer class for Bundle Adjustment (API of the reference's class of the same name)."""

    default_conf = {
        "depth_loss_name": "cauchy",
        "ref3d_loss_name": "trivial",
        "reproj_loss_name": "SOFT_L1",
        "reproj_loss_scale": 1.5,
        "scale_filter": True,
        "scale_filter_factor": 1.5,
        "metric_scale_filter": True,
        "rob_std": 2,
        "truncation_mode": "mad",
        "gross_outliers": False,
        "single_rescale": True,
        "min_truncation_mult": None,
        "verbose": 0,
    }

    def _init(self, mpsfm_rec, correspondences=None, backend=None):
        self.mpsfm_rec = mpsfm_rec
        self.correspondences = correspondences
        self.truncation_multiplier = 1
        self.get_loss = {"trivial": LOSS_TRIVIAL, "cauchy": LOSS_CAUCHY, "softl1": LOSS_SOFT_L1}
        self.backend = backend if backend is not None else HipBackend()
        self.last_summary = None

    # ------------------------------------------------------------------------------------------
    def _yield_problem_parameters(self, optim_ids, proj_depths=False):
        """reference :50-65"""
        for imid in optim_ids:
            image = self.mpsfm_rec.images[imid]
            camera = self.mpsfm_rec.rec.cameras[image.camera_id]
            pt2D_ids = image.get_observation_point2D_idxs()
            kps_with3D = image.keypoint_coordinates(pt2D_ids)
            p3d_ids = image.point3D_ids(pt2D_ids)
            kwargs = {"imid": imid, "image": image, "camera": camera, "pt3D_ids": p3d_ids, "kps": kps_with3D}
            kwargs["obsdepths"] = image.depth.data_prior_at_kps(kps_with3D)
            kwargs["valid"] = image.depth.valid_at_kps(kps_with3D)
            if proj_depths:
                _, _, _, depth3d, _ = self.mpsfm_rec.project_image_3d_points(imid, kwargs["pt3D_ids"])
                kwargs["projdepths"] = depth3d
            yield kwargs

    def _gather_reprojection(self, optim_ids, variable_points, fix_pose):
        """The residual blocks pycolmap.create_default_bundle_adjuster adds (reference :85-104):
        every observation of every bundle image; for explicitly variable points also their
        observations in images outside the bundle, with those poses constant; a point whose track
        is not completely inside the problem is constant (COLMAP ParameterizePoints)."""
        rec = self.mpsfm_rec
        image_ids = list(optim_ids)
        cam_of = {imid: i for i, imid in enumerate(image_ids)}
        in_config = set(image_ids)
        # vectorised over each image's observations (no per-observation Python work)
        cams_l, pids_l, xy_l = [], [], []
        for imid in image_ids:
            image = rec.images[imid]
            p2d = np.asarray(image.get_observation_point2D_idxs(), dtype=np.int64)
            if len(p2d) == 0:
                continue
            cams_l.append(np.full(len(p2d), cam_of[imid], np.int32))
            pids_l.append(np.asarray(image.point3D_ids(p2d), dtype=np.uint64))
            xy_l.append(np.asarray(image.keypoint_coordinates(p2d), dtype=np.float64).reshape(-1, 2))
        if pids_l:
            uniq, inv, counts = np.unique(np.concatenate(pids_l), return_inverse=True, return_counts=True)
            obs_cam, obs_pt, obs_xy = list(cams_l), [inv.astype(np.int32)], list(xy_l)
            obs_cam = [np.concatenate(obs_cam)]
        else:
            uniq, counts = np.zeros(0, np.uint64), np.zeros(0, np.int64)
            obs_cam, obs_pt, obs_xy = [], [], []
        point_ids = [int(v) for v in uniq]
        pt_of = dict(zip(point_ids, range(len(point_ids))))
        num_obs = dict(zip(point_ids, (int(c) for c in counts)))
        extra_cam, extra_pt, extra_xy = [], [], []
        for pid in variable_points:
            point = rec.points3D[pid]
            if num_obs.get(pid, 0) == point.track.length():
                continue
            if pid not in pt_of:
                pt_of[pid] = len(point_ids)
                point_ids.append(pid)
            for el in point.track.elements:
                if el.image_id in in_config:
                    continue
                if el.image_id not in cam_of:
                    cam_of[el.image_id] = len(image_ids)
                    image_ids.append(el.image_id)
                extra_cam.append(cam_of[el.image_id]); extra_pt.append(pt_of[pid])
                extra_xy.append(np.asarray(rec.images[el.image_id].points2D[el.point2D_idx].xy, np.float64))
                num_obs[pid] = num_obs.get(pid, 0) + 1
        if extra_cam:
            obs_cam.append(np.array(extra_cam, np.int32)); obs_pt.append(np.array(extra_pt, np.int32))
            obs_xy.append(np.array(extra_xy, np.float64).reshape(-1, 2))
        obs_cam = np.concatenate(obs_cam) if obs_cam else np.zeros(0, np.int32)
        obs_pt = np.concatenate(obs_pt) if obs_pt else np.zeros(0, np.int32)
        obs_xy = np.concatenate(obs_xy) if obs_xy else np.zeros((0, 2))
        self._sorted_point_ids = uniq  # bundle-image points, sorted: index = position (depth blocks look ids up here)
        n_cfg = len(optim_ids)
        pose_const = np.ones(len(image_ids), np.uint8)
        for ii in range(n_cfg):
            pose_const[ii] = 1 if (fix_pose or ii == 0) else 0
        pt_const = np.array([1 if rec.points3D[pid].track.length() > num_obs[pid] else 0 for pid in point_ids], np.uint8)
        gauge = 1 if (not fix_pose and n_cfg > 1) else -1
        return image_ids, cam_of, point_ids, pt_of, obs_cam, obs_pt, obs_xy, pose_const, pt_const, gauge

    def _build_problem(self, bundle, fix_pose, fix_scale, mode=None, depth_loss_name=None, allow_scale_filter=False,
                       param_multiplier=1, depth_type="update", solve=True, **kw):
        """reference __build_problem :67-185"""
        conf, rec = self.conf, self.mpsfm_rec
        optim_ids = list(bundle["optim_ids"])
        depth_loss_name = depth_loss_name or conf.depth_loss_name
        depth_loss_type = self.get_loss[depth_loss_name]
        shift_scale = {imid: np.array([0.0, 0.0]) for imid in optim_ids}

        variable_points = []
        if mode == "local":
            variable_points = [p for p in bundle["pts3D"] if rec.points3D[p].track.length() < 15]
        (image_ids, cam_of, point_ids, pt_of, obs_cam, obs_pt, obs_xy, pose_const, pt_const,
         gauge) = self._gather_reprojection(optim_ids, variable_points, fix_pose)

        kp_std = float(np.median([rec.images[imid].kp_std for imid in optim_ids]))
        scale_filter_factor = conf.scale_filter_factor
        gross_outliers = conf.gross_outliers
        param_multiplier = param_multiplier * self.truncation_multiplier

        dobs_cam, dobs_pt, dobs_d, dobs_m, dobs_a = [], [], [], [], []
        for imid in optim_ids:
            image = rec.images[imid]
            if not image.depth.activated:
                continue
            p2Ds = np.asarray(image.get_observation_point2D_idxs(), dtype=np.int64)
            if len(p2Ds) == 0:
                continue
            kps = np.asarray(image.keypoint_coordinates(p2Ds))
            valid = image.depth.valid_at_kps(kps)
            kps = kps[valid]
            depths = image.depth.data_at_kps(kps) if depth_type == "update" else image.depth.data_prior_at_kps(kps)
            p2Ds = p2Ds[valid]
            p3Ds = np.asarray(image.point3D_ids(p2Ds), dtype=np.uint64)
            if len(p3Ds) == 0:
                continue
            _, _, _, depth3d, _ = rec.project_image_3d_points(imid, p3Ds)
            mask = depths > 0
            if allow_scale_filter and conf.scale_filter:
                div = depths / depth3d
                mask = mask & (div < scale_filter_factor) & (div > (1 / scale_filter_factor))
            uu = image.depth.uncertainty_update
            variances = np.asarray(uu, np.float64)[p2Ds] if isinstance(uu, np.ndarray) else np.array([uu[int(i)] for i in p2Ds], np.float64)
            if gross_outliers and image.depth.activated:
                whitened = np.abs(np.log(depths).clip(1e-6, None) - np.log(depth3d).clip(1e-6, None)) / variances**0.5
                mask = mask & (whitened < 3)
            if np.sum(mask) == 0:
                self.log("No valid points for depth regularizing", level=1)
                continue
            depths, variances, p3Ds = depths[mask], variances[mask], p3Ds[mask]
            inv_uncert = 1 / variances.clip(1e-6, None)
            m = param_multiplier * conf.rob_std
            params = m * variances**0.5 / depths
            magnitudes = depths**2 * inv_uncert
            dobs_cam.append(np.full(len(p3Ds), cam_of[imid], np.int32))
            dobs_pt.append(np.searchsorted(self._sorted_point_ids, p3Ds).astype(np.int32))
            dobs_d.append(depths); dobs_m.append(magnitudes); dobs_a.append(params)

        n_cams = len(image_ids)
        cam_ids = [rec.images[i].camera_id for i in image_ids]
        uniq = sorted(set(cam_ids))
        intr = np.array([pinhole_params(rec.rec.cameras[c]) for c in uniq]).reshape(-1, 4)
        prob = BAProblem(
            cam_quat=np.array([rec.images[i].cam_from_world.rotation.quat for i in image_ids]).reshape(-1, 4),
            cam_t=np.array([rec.images[i].cam_from_world.translation for i in image_ids]).reshape(-1, 3),
            pts=np.array([rec.points3D[p].xyz for p in point_ids]).reshape(-1, 3),
            cam_intr=intr, cam_intr_idx=np.array([uniq.index(c) for c in cam_ids], np.int32),
            pose_const=pose_const, pt_const=pt_const,
            obs_cam=np.array(obs_cam, np.int32), obs_pt=np.array(obs_pt, np.int32),
            obs_xy=np.array(obs_xy, np.float64).reshape(-1, 2), gauge_axis_cam=gauge if n_cams > 1 else -1,
            reproj_loss_type=_COLMAP_LOSS[str(conf.reproj_loss_name).upper()],
            reproj_loss_scale=conf.reproj_loss_scale * kp_std, reproj_loss_magnitude=1 / kp_std**2,
            dobs_cam=np.concatenate(dobs_cam) if dobs_cam else np.zeros(0, np.int32),
            dobs_pt=np.concatenate(dobs_pt) if dobs_pt else np.zeros(0, np.int32),
            dobs_depth=np.concatenate(dobs_d) if dobs_d else np.zeros(0),
            dobs_magnitude=np.concatenate(dobs_m) if dobs_m else np.zeros(0),
            dobs_param=np.concatenate(dobs_a) if dobs_a else np.zeros(0),
            depth_loss_type=depth_loss_type,
        )
        flat = FlatProblem(prob, image_ids, point_ids, shift_scale)
        if solve:
            self.solve(flat)
        return flat, shift_scale

    # ------------------------------------------------------------------------------------------
    def solve(self, problem: FlatProblem):
        """Solves the optimization problem (reference :285-293) and writes the result back in place."""
        summary = self.backend.solve(problem.prob)
        problem.summary = summary
        self.last_summary = summary
        rec, prob = self.mpsfm_rec, problem.prob
        for ci, imid in enumerate(problem.image_ids):
            if prob.pose_const[ci]:
                continue
            pose = rec.images[imid].cam_from_world
            pose.rotation.quat[:] = prob.cam_quat[ci]
            pose.translation[:] = prob.cam_t[ci]
        for pi, pid in enumerate(problem.point_ids):
            if not prob.pt_const[pi]:
                rec.points3D[pid].xyz[:] = prob.pts[pi]
        self.log(
            f"LM iterations {summary['num_iterations']}, cost {summary['initial_cost']:.6e} -> {summary['final_cost']:.6e}, "
            f"{summary['termination']}", level=2)

    def ba(self, bundle, mode, **kwargs):
        """Optimizes per frame data and 3d points in entire reconstruction (reference :263-266)."""
        problem, _ = self._build_problem(bundle, fix_pose=False, fix_scale=True, mode=mode, **kwargs)
        return problem, True

    def refine_3d_points(self, bundle, **kwargs):
        """Refines triangulated 3d points with depth maps keeping poses fixed (reference :276-283)."""
        problem, _ = self._build_problem(bundle, fix_pose=True, fix_scale=True,
                                         depth_loss_name=self.conf.ref3d_loss_name, **kwargs)
        return problem, True

    def calculate_point_covs(self, bundle):
        """Calculates point covariances for the given bundle (reference :244-261): reprojection-only
        problem, trivial loss with magnitude 1/kp_std^2, every bundle point variable (which pulls in
        its observations outside the bundle), covariance of each point with all else constant."""
        rec = self.mpsfm_rec
        optim_ids = list(bundle["optim_ids"])
        (image_ids, _, point_ids, pt_of, obs_cam, obs_pt, obs_xy, pose_const, pt_const,
         _) = self._gather_reprojection(optim_ids, list(bundle["pts3D"]), fix_pose=True)
        kp_std = float(np.median([rec.images[imid].kp_std for imid in optim_ids]))
        cam_ids = [rec.images[i].camera_id for i in image_ids]
        uniq = sorted(set(cam_ids))
        prob = BAProblem(
            cam_quat=np.array([rec.images[i].cam_from_world.rotation.quat for i in image_ids]).reshape(-1, 4),
            cam_t=np.array([rec.images[i].cam_from_world.translation for i in image_ids]).reshape(-1, 3),
            pts=np.array([rec.points3D[p].xyz for p in point_ids]).reshape(-1, 3),
            cam_intr=np.array([pinhole_params(rec.rec.cameras[c]) for c in uniq]).reshape(-1, 4),
            cam_intr_idx=np.array([uniq.index(c) for c in cam_ids], np.int32), pose_const=pose_const, pt_const=pt_const,
            obs_cam=np.array(obs_cam, np.int32), obs_pt=np.array(obs_pt, np.int32),
            obs_xy=np.array(obs_xy, np.float64).reshape(-1, 2), reproj_loss_type=LOSS_TRIVIAL,
            reproj_loss_scale=1.0, reproj_loss_magnitude=1 / kp_std**2,
        )
        covs = self.backend.point_covs(prob)
        for p3Did in bundle["pts3D"]:
            if p3Did in pt_of:
                rec.point_covs.data[p3Did] = covs[pt_of[p3Did]]

    # ------------------------------------------------------------------------------------------
    def _build_shiftscale_problem(self, bundle, allow_scale_filter=False, allow_metric_scale_filter=False):
        """Per-image median log-scale of projected vs prior depth (reference :187-242; no Ceres)."""
        conf, rec = self.conf, self.mpsfm_rec
        shift_scale = {}
        scale_filter, factor = conf.scale_filter, conf.scale_filter_factor
        metric = conf.metric_scale_filter
        single = conf.single_rescale
        for kw in self._yield_problem_parameters(bundle["optim_ids"], proj_depths=scale_filter or metric):
            imid, p3dids = kw["imid"], kw["pt3D_ids"]
            pose = kw["image"].cam_from_world
            if (factor or metric) and ("ref_id" in bundle and imid != bundle["ref_id"] and single):
                continue
            valid = np.array(kw["valid"], dtype=bool)
            if allow_metric_scale_filter and metric and ((imid == bundle["ref_id"]) or (not single)):
                scale = kw["projdepths"] / (kw["obsdepths"].clip(1e-6, None))
                proposed = scale * rec.images[imid].depth.scale
                map_scale = np.mean([rec.images[i].depth.scale for i in bundle["optim_ids"] if i != imid])
                div = map_scale / proposed
                valid = valid & (div < 1.5) & (div > (1 / 1.5))
                if valid.sum() == 0:
                    self.log("WARNING: all points are outliers for the metric scale fit; using the map scale", level=0)
                    shift_scale[imid] = np.array([0.0, np.log(map_scale / rec.images[imid].depth.scale)])
                    return shift_scale, True
            if allow_scale_filter and scale_filter and not allow_metric_scale_filter:
                div = kw["obsdepths"] / kw["projdepths"]
                valid = valid & (div < factor) & (div > (1 / factor))
            z = (pose * rec.point3D_coordinates(p3dids))[:, -1][valid]
            odepth = kw["obsdepths"][valid]
            shift_scale[imid] = np.array([0.0, np.median(np.log((z / odepth).clip(1e-6, None)))])
        return shift_scale, True

    def optimize_prior_shiftscale(self, bundle, **kwargs):
        """{imid: (shift, scale)} (reference :268-274)."""
        shift_scale, success = self._build_shiftscale_problem(bundle, **kwargs)
        if not success:
            return None, False
        return {imid: (shift, np.exp(scale)) for imid, (shift, scale) in shift_scale.items()}, True

    def update_truncation_multiplier(self, imids):
        """sigma of the whitened log-depth errors by MAD over the given images (reference :295-333)."""
        rec = self.mpsfm_rec
        D, D3d, stds = [], [], []
        for imid in imids:
            image = rec.images[imid]
            p2Ds = np.asarray(image.get_observation_point2D_idxs(), dtype=np.int64)
            if len(p2Ds) == 0:
                continue
            kps = np.asarray(image.keypoint_coordinates(p2Ds))
            valid = image.depth.valid_at_kps(kps)
            depths = image.depth.data_at_kps(kps[valid])
            p2Ds = p2Ds[valid]
            p3Ds = np.asarray(image.point3D_ids(p2Ds), dtype=np.uint64)
            mask = depths > 0
            if mask.sum() == 0:
                continue
            _, _, _, depth3d, _ = rec.project_image_3d_points(imid, p3Ds[mask])
            uu = image.depth.uncertainty_update
            D.append(depths[mask]); D3d.append(depth3d)
            vv = np.asarray(uu, np.float64)[p2Ds[mask]] if isinstance(uu, np.ndarray) else np.array([uu[int(i)] for i in p2Ds[mask]])
            stds.append(vv**0.5)
        depths, depth3ds, dstds = np.concatenate(D), np.concatenate(D3d), np.concatenate(stds)
        log_stds = np.clip(dstds / depths, 1e-6, None)
        _, sigma = fit_robust_gaussian_mad((np.log(depths) - np.log(depth3ds)) / log_stds)
        self.truncation_multiplier = sigma
        if self.conf.min_truncation_mult is not None:
            self.truncation_multiplier = max(self.truncation_multiplier, self.conf.min_truncation_mult)
