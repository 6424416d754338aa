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
from ...utils.ids import index_in_sorted, unique_ids
from ...problem import LOSS_BY_NAME, LOSS_CAUCHY, LOSS_SOFT_L1, LOSS_TRIVIAL, BAProblem
from ..scene.prior_gather import F_GROSS, F_POSITIVE, F_SCALE, F_VALID, gather_bundle
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

    def depth_blocks(self, **kw) -> dict:
        from ... import capi

        return capi.depth_blocks(device=self.device, **kw)


class _SortedIndex:
    """id -> position in a sorted id array with the mapping interface the callers use (`in`, [])."""

    def __init__(self, ids):
        self.ids = ids

    def _pos(self, pid):
        i = int(np.searchsorted(self.ids, pid))
        return i if i < len(self.ids) and self.ids[i] == pid else -1

    def __contains__(self, pid):
        return self._pos(pid) >= 0

    def __getitem__(self, pid):
        i = self._pos(pid)
        if i < 0:
            raise KeyError(pid)
        return i


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
            uniq, inv, counts = unique_ids(np.concatenate(pids_l), return_counts=True)
            obs_cam, obs_pt, obs_xy = [np.concatenate(cams_l)], [inv.astype(np.int32)], list(xy_l)
        else:
            uniq, counts = np.zeros(0, np.uint64), np.zeros(0, np.int64)
            obs_cam, obs_pt, obs_xy = [], [], []
        self._sorted_point_ids = uniq  # bundle-image points, sorted: index = position (depth blocks look ids up here)
        point_ids = uniq.astype(np.int64)
        num_obs = counts.astype(np.int64)
        # explicitly variable points whose track leaves the bundle: their outside observations come along, constant poses
        vp = np.fromiter((int(p) for p in variable_points), np.int64, len(variable_points)) if len(variable_points) else np.zeros(0, np.int64)
        if len(vp):
            pos = np.searchsorted(point_ids, vp)
            inside = (pos < len(point_ids)) & (point_ids[np.minimum(pos, max(len(point_ids) - 1, 0))] == vp) if len(point_ids) else np.zeros(len(vp), bool)
            n_in = np.where(inside, num_obs[np.minimum(pos, max(len(point_ids) - 1, 0))] if len(point_ids) else 0, 0)
            todo = vp[self._track_lengths(vp) != n_in]
        else:
            todo = vp
        pt_of = None
        if len(todo):
            # their observations in images OUTSIDE the bundle, as (image id, point id, xy) in a canonical order (point, image):
            # from the image side with bulk accessors when that is cheaper than walking every track in Python
            out_img, out_pid, out_xy = self._outside_observations(np.asarray(todo, np.int64), in_config)
            pos = np.searchsorted(point_ids, out_pid) if len(point_ids) else np.zeros(len(out_pid), np.int64)
            known = (pos < len(point_ids)) & (point_ids[np.minimum(pos, max(len(point_ids) - 1, 0))] == out_pid) if len(point_ids) else np.zeros(len(out_pid), bool)
            todo_in = np.isin(todo, point_ids) if len(point_ids) else np.zeros(len(todo), bool)
            extra_ids = np.asarray(todo, np.int64)[~todo_in]  # explicit points no bundle image observes: appended behind the sorted ones
            n_sorted = len(point_ids)
            extra_pos = {int(v): n_sorted + i for i, v in enumerate(extra_ids)}
            ept = np.where(known, pos, 0).astype(np.int64)
            if (~known).any():
                ept[~known] = np.fromiter((extra_pos[int(v)] for v in out_pid[~known]), np.int64, int((~known).sum()))
            out_images = np.unique(out_img)  # outside images in ascending id order behind the bundle's
            first_out = len(image_ids)
            for imid in out_images:
                cam_of[int(imid)] = len(image_ids)
                image_ids.append(int(imid))
            ecam = (first_out + np.searchsorted(out_images, out_img)).astype(np.int32)
            if len(extra_ids):
                point_ids = np.concatenate([point_ids, extra_ids])
                num_obs = np.concatenate([num_obs, np.zeros(len(extra_ids), np.int64)])
                pt_of = {int(v): i for i, v in enumerate(point_ids)}
            if len(ecam):
                np.add.at(num_obs, ept, 1)
                obs_cam.append(ecam); obs_pt.append(ept.astype(np.int32)); obs_xy.append(out_xy.reshape(-1, 2))
        obs_cam = np.concatenate(obs_cam) if obs_cam else np.zeros(0, np.int32)
        obs_pt = np.concatenate(obs_pt) if obs_pt else np.zeros(0, np.int32)
        obs_xy = np.concatenate(obs_xy) if obs_xy else np.zeros((0, 2))
        n_cfg = len(optim_ids)
        pose_const = np.ones(len(image_ids), np.uint8)
        for ii in range(n_cfg):
            pose_const[ii] = 1 if (fix_pose or ii == 0) else 0
        pt_const = (self._track_lengths(point_ids) > num_obs).astype(np.uint8)
        if pt_of is None:
            pt_of = _SortedIndex(point_ids)
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
            cand = np.fromiter((int(p) for p in bundle["pts3D"]), np.int64, len(bundle["pts3D"]))
            variable_points = cand[self._track_lengths(cand) < 15] if len(cand) else cand
        (image_ids, cam_of, point_ids, pt_of, obs_cam, obs_pt, obs_xy, pose_const, pt_const,
         gauge) = self._gather_reprojection(optim_ids, variable_points, fix_pose)

        kp_std = float(np.median([rec.images[imid].kp_std for imid in optim_ids]))
        scale_filter_factor = conf.scale_filter_factor
        gross_outliers = conf.gross_outliers
        param_multiplier = param_multiplier * self.truncation_multiplier

        # depth blocks of every activated bundle image (reference :124-176): the keypoints are gathered
        # per image through the reference's accessors, the sampling / projection / masks / weights run in ONE launch
        dobs_cam = dobs_pt = np.zeros(0, np.int32)
        dobs_d = dobs_m = dobs_a = np.zeros(0)
        g = gather_bundle(rec, optim_ids, depth_type)
        if g is not None:
            out = self.backend.depth_blocks(
                depth_maps=g["depth_maps"], valid_maps=g["valid_maps"], sx=g["sx"], sy=g["sy"], cam_quat=g["cam_quat"], cam_t=g["cam_t"],
                obs_img=g["obs_img"], obs_xy=g["obs_xy"], obs_var=g["obs_var"], obs_pt=g["obs_pt"], pts=self._coordinates(g["point_ids"]),
                scale_filter_factor=scale_filter_factor, multiplier=param_multiplier * conf.rob_std)
            f = out["flags"]
            mask = ((f & F_VALID) != 0) & ((f & F_POSITIVE) != 0)
            if allow_scale_filter and conf.scale_filter:
                mask &= (f & F_SCALE) != 0
            if gross_outliers:
                mask &= (f & F_GROSS) != 0
            for _ in np.flatnonzero(np.bincount(g["obs_img"][mask], minlength=len(g["images"])) == 0):
                self.log("No valid points for depth regularizing", level=1)
            cam_idx = np.array([cam_of[i] for i in g["images"]], np.int32)
            dobs_cam = cam_idx[g["obs_img"][mask]]
            dobs_pt = index_in_sorted(self._sorted_point_ids, g["obs_pid"][mask]).astype(np.int32)
            dobs_d, dobs_m, dobs_a = out["depth"][mask], out["magnitude"][mask], out["param"][mask]

        n_cams = len(image_ids)
        cam_ids = [rec.images[i].camera_id for i in image_ids]
        uniq = sorted(set(cam_ids))
        intr = np.array([pinhole_params(rec.rec.cameras[c]) for c in uniq]).reshape(-1, 4)
        prob = BAProblem(
            cam_quat=np.array([rec.images[i].cam_from_world.rotation.quat for i in image_ids]).reshape(-1, 4),
            cam_t=np.array([rec.images[i].cam_from_world.translation for i in image_ids]).reshape(-1, 3),
            pts=self._coordinates(point_ids),
            cam_intr=intr, cam_intr_idx=np.array([uniq.index(c) for c in cam_ids], np.int32),
            pose_const=pose_const, pt_const=pt_const,
            obs_cam=np.array(obs_cam, np.int32), obs_pt=np.array(obs_pt, np.int32),
            obs_xy=np.array(obs_xy, np.float64).reshape(-1, 2), gauge_axis_cam=gauge if n_cams > 1 else -1,
            reproj_loss_type=_COLMAP_LOSS[str(conf.reproj_loss_name).upper()],
            reproj_loss_scale=conf.reproj_loss_scale * kp_std, reproj_loss_magnitude=1 / kp_std**2,
            dobs_cam=dobs_cam, dobs_pt=dobs_pt, dobs_depth=dobs_d, dobs_magnitude=dobs_m, dobs_param=dobs_a,
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
        var = np.flatnonzero(prob.pt_const == 0)
        ids = np.asarray(problem.point_ids)[var]
        if hasattr(rec, "set_point3D_coordinates"):  # bulk write when the scene offers one
            rec.set_point3D_coordinates(ids, prob.pts[var])
        else:
            for pi, pid in zip(var, ids):
                rec.points3D[int(pid)].xyz[:] = prob.pts[pi]
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
            pts=self._coordinates(point_ids),
            cam_intr=np.array([pinhole_params(rec.rec.cameras[c]) for c in uniq]).reshape(-1, 4),
            cam_intr_idx=np.array([uniq.index(c) for c in cam_ids], np.int32), pose_const=pose_const, pt_const=pt_const,
            obs_cam=np.array(obs_cam, np.int32), obs_pt=np.array(obs_pt, np.int32),
            obs_xy=np.array(obs_xy, np.float64).reshape(-1, 2), reproj_loss_type=LOSS_TRIVIAL,
            reproj_loss_scale=1.0, reproj_loss_magnitude=1 / kp_std**2,
        )
        covs = self.backend.point_covs(prob)
        data = rec.point_covs.data
        index = {int(v): i for i, v in enumerate(point_ids)}
        for p3Did in bundle["pts3D"]:
            i = index.get(p3Did)
            if i is not None:
                data[p3Did] = covs[i]

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
        """sigma of the whitened log-depth errors by MAD over the given images (reference :295-333); the per-keypoint
        sampling, projection and whitening run in the same launch as the depth-block selection."""
        g = gather_bundle(self.mpsfm_rec, imids, "update")
        out = self.backend.depth_blocks(
            depth_maps=g["depth_maps"], valid_maps=g["valid_maps"], sx=g["sx"], sy=g["sy"], cam_quat=g["cam_quat"], cam_t=g["cam_t"],
            obs_img=g["obs_img"], obs_xy=g["obs_xy"], obs_var=g["obs_var"], obs_pt=g["obs_pt"], pts=self._coordinates(g["point_ids"]),
            scale_filter_factor=self.conf.scale_filter_factor, multiplier=1.0)
        f = out["flags"]
        sel = ((f & F_VALID) != 0) & ((f & F_POSITIVE) != 0)
        _, sigma = fit_robust_gaussian_mad(out["whitened"][sel])
        self.truncation_multiplier = sigma
        if self.conf.min_truncation_mult is not None:
            self.truncation_multiplier = max(self.truncation_multiplier, self.conf.min_truncation_mult)

    def _outside_observations(self, todo, in_config):
        """Observations of the points `todo` in images outside `in_config`: (image ids, point ids, xy), ordered by (point, image).
        Two ways to the same arrays: walking point.track.elements (what pycolmap's C++ bundle adjuster does for
        config.add_variable_point, reference :88-91) costs one Python step per track element; sweeping the other images'
        observation lists with the bulk accessors costs a few NumPy calls per image.  The cheaper one by a simple model is taken."""
        rec = self.mpsfm_rec
        todo_sorted = np.sort(todo)
        others = [(imid, im) for imid, im in rec.images.items() if imid not in in_config]
        n_elements = int(self._track_lengths(todo).sum())
        n_cfg_obs = 0
        for imid in list(in_config)[:4]:
            n_cfg_obs += len(rec.images[imid].get_observation_point2D_idxs())
        per_image = n_cfg_obs / max(min(len(in_config), 4), 1)
        cost_tracks = 1.0 * n_elements                               # ~1 us per element visited from Python
        cost_images = len(others) * (25.0 + 0.03 * per_image)        # ~25 us of calls + ~30 ns per observation and image
        img_l, pid_l, xy_l = [], [], []
        force = getattr(self, "outside_method", None)  # tests: "images" / "tracks"
        if force == "images" or (force is None and cost_images <= cost_tracks):
            for imid, image in others:
                p2d = np.asarray(image.get_observation_point2D_idxs(), dtype=np.int64)
                if len(p2d) == 0:
                    continue
                pids = np.asarray(image.point3D_ids(p2d), dtype=np.uint64).astype(np.int64)
                pos = np.minimum(np.searchsorted(todo_sorted, pids), len(todo_sorted) - 1)
                hit = todo_sorted[pos] == pids
                if not hit.any():
                    continue
                img_l.append(np.full(int(hit.sum()), imid, np.int64))
                pid_l.append(pids[hit])
                xy_l.append(np.asarray(image.keypoint_coordinates(p2d[hit]), dtype=np.float64).reshape(-1, 2))
        else:
            for pid in todo:
                els = [(el.image_id, el.point2D_idx) for el in rec.points3D[int(pid)].track.elements if el.image_id not in in_config]
                if not els:
                    continue
                img_l.append(np.array([e[0] for e in els], np.int64))
                pid_l.append(np.full(len(els), int(pid), np.int64))
                xy_l.append(np.array([np.asarray(rec.images[i].points2D[j].xy, np.float64) for i, j in els], np.float64).reshape(-1, 2))
        if not img_l:
            return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 2))
        img, pid, xy = np.concatenate(img_l), np.concatenate(pid_l), np.concatenate(xy_l)
        order = np.lexsort((img, pid))
        return img[order], pid[order], xy[order]

    def _track_lengths(self, point_ids):
        """track length of every given point: a bulk accessor when the scene has one, else the reference's per-point
        point3D.track.length()"""
        rec = self.mpsfm_rec
        if hasattr(rec, "point3D_track_lengths"):
            return np.asarray(rec.point3D_track_lengths(point_ids), np.int64)
        return np.array([rec.points3D[int(p)].track.length() for p in point_ids], np.int64)

    def _coordinates(self, point_ids):
        """xyz of the given points: the reference's bulk accessor mpsfm_rec.point3D_coordinates (used at :233)."""
        return np.asarray(self.mpsfm_rec.point3D_coordinates(point_ids), np.float64).reshape(-1, 3)
