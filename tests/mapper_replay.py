"""TEST SCAFFOLDING: the call ORDER of the reference's MpsfmMapper around the hot path, replayed on the NumPy scene.

The mapper itself is out of scope and stays the reference's; what the build owes it is the
contract of SURVEY row a-13: that Optimizer / MpsfmTriangulator / ObservationManager / Image.integrate can be called
in the reference's order with the reference's arguments.  Each method below names the reference lines whose sequence
of calls it follows (mpsfm/sfm/mapper/base.py); no arithmetic lives here.
"""

from __future__ import annotations

import numpy as np


class MapperReplay:
    # COLMAP defaults the mapper inherits and the overrides at mapper/base.py:35-40
    filter_max_reproj_error = 4.0
    filter_min_tri_angle = 0.001
    local_ba_num_images = 6

    def __init__(self, mpsfm_rec, optimizer, triangulator=None, integrable=None, integrate=True, int_covs=False):
        self.mpsfm_rec, self.optimizer, self.triangulator = mpsfm_rec, optimizer, triangulator
        self.integrable = integrable or {}  # imid -> object with .integrate() (Image in the reference)
        self.integrate, self.int_covs = integrate, int_covs
        self.first_refinement = True
        self.calls = []  # the order of hot-path calls, for the contract assertions

    # -- base.py:619-631 ----------------------------------------------------------------------------------------
    def integrate_bundle(self, imids, int_covs=True, cache_device="cpu", **kwargs):
        for imid in imids:
            self.calls.append(("integrate", imid))
            self.integrable[imid].integrate(cache_device=cache_device)
        self.first_refinement = False
        return True

    # -- base.py:633-654 ----------------------------------------------------------------------------------------
    def adjust_bundle(self, bundle, int_covs, mode="global", refimid=None, allow_scale_filter=False, **kwargs):
        if self.integrate:
            integrate_imids = bundle["optim_ids"] if mode == "global" else [refimid]
            if not self.integrate_bundle(integrate_imids, int_covs, cache_device="cuda" if mode == "local" else "cpu", **kwargs):
                return None, False
        if mode == "global":
            self.calls.append(("update_truncation_multiplier",))
            self.optimizer.update_truncation_multiplier(self.mpsfm_rec.reg_image_ids())
        self.calls.append(("ba", mode))
        problem, success = self.optimizer.ba(bundle, mode=mode, allow_scale_filter=allow_scale_filter, **kwargs)
        return (problem, True) if success else (None, False)

    # -- base.py:420-440 ----------------------------------------------------------------------------------------
    def _refinement(self, bundle, int_covs, mode="global", refimid=None, allow_scale_filter=False, **kwargs):
        _, success = self.adjust_bundle(bundle, int_covs, mode=mode, refimid=refimid, allow_scale_filter=allow_scale_filter, **kwargs)
        if not success:
            return None, False
        num_observations = len(bundle["pts3D"])
        num_changed, filtered_imids = self.filter_bundle(bundle)
        if self.triangulator is not None:
            self.calls.append(("complete_and_merge_tracks",))
            num_changed += self.triangulator.complete_and_merge_tracks(bundle["pts3D"])
        changed = 0 if num_observations == 0 else num_changed / num_observations
        if len(filtered_imids) > 0:
            return "deregistered", False
        return changed, True

    # -- base.py:516-539 ----------------------------------------------------------------------------------------
    def post_init_refinement(self):
        self.first_refinement = True
        bundle = self.find_global_bundle()
        self.calls.append(("calculate_point_covs",))
        self.optimizer.calculate_point_covs(bundle)
        self.calls.append(("optimize_prior_shiftscale",))
        shift_scale, success = self.optimizer.optimize_prior_shiftscale(bundle)
        if not success:
            return False
        self.mpsfm_rec.rescale_all(shift_scale)
        self.mpsfm_rec.activate_depths(bundle["optim_ids"])
        self.calls.append(("refine_3d_points",))
        if not self.optimizer.refine_3d_points(bundle):
            return False
        self.filter_all()
        return bool(self.mpsfm_rec.registered_images)

    # -- base.py:541-617 (depth-consistency branch off: that checker is outside the hot path) --------------------
    def post_registration_refinement(self, imid):
        self.first_refinement = True
        rec = self.mpsfm_rec
        if rec.images[imid].depth.activated:
            rec.images[imid].depth.reset()
        local_bundle = self.find_local_bundle(imid)
        _, filtered = self.filter_bundle(local_bundle)
        if imid in filtered:
            return False
        self.calls.append(("refine_3d_points", "pre"))
        _, success = self.optimizer.refine_3d_points(local_bundle, depth_type="prior" if not self.integrate else "update")
        if not success:
            return False
        local_bundle = self.find_local_bundle(imid)
        _, filtered = self.filter_bundle(local_bundle)
        if len(filtered) > 0:
            return False
        observed_bundle = self.find_subset_bundle(local_bundle)
        self.calls.append(("calculate_point_covs",))
        self.optimizer.calculate_point_covs(observed_bundle)
        self.calls.append(("optimize_prior_shiftscale",))
        shift_scale, success = self.optimizer.optimize_prior_shiftscale(local_bundle, allow_metric_scale_filter=True)
        if not success:
            return False
        rec.rescale_all(shift_scale)
        rec.activate_depths({imid})
        if self.integrate and not self.integrate_bundle([imid], int_covs=self.int_covs):
            return False
        self.calls.append(("refine_3d_points", "post"))
        if not self.optimizer.refine_3d_points(local_bundle, depth_type="prior" if not self.integrate else "update"):
            return False
        local_bundle = self.find_local_bundle(imid)
        self.filter_bundle(local_bundle)
        return imid in rec.registered_images

    # -- base.py:442-474 ----------------------------------------------------------------------------------------
    def iterative_local_refinement(self, imid, max_refinements=2, max_change=0.001):
        if self.triangulator is not None:
            self.triangulator.complete_and_merge_all_tracks()
        for _ in range(max_refinements):
            local_bundle = self.find_local_bundle(imid)
            observed_bundle = self.find_subset_bundle(local_bundle)
            self.calls.append(("calculate_point_covs",))
            self.optimizer.calculate_point_covs(observed_bundle)
            changed, success = self._refinement(local_bundle, self.int_covs, mode="local", refimid=imid, allow_scale_filter=True)
            if not success:
                return False
            if changed < max_change:
                break
        return True

    # -- filters: base.py:686-711, 766-797 ------------------------------------------------------------------------
    def _max_err(self):
        return self.filter_max_reproj_error * np.median([im.kp_std for im in self.mpsfm_rec.images.values()])

    def filter_all(self):
        rec = self.mpsfm_rec
        self.calls.append(("filter_all",))
        rec.obs.filter_observations_with_negative_depth()
        n = rec.obs.filter_all_points3D(self._max_err(), self.filter_min_tri_angle)
        return n, self.filter_images()

    def filter_bundle(self, bundle, filter_ims=True):
        rec = self.mpsfm_rec
        self.calls.append(("filter_bundle",))
        rec.obs.filter_observations_with_negative_depth()
        n = self.filter_local_points3D(bundle, self._max_err(), self.filter_min_tri_angle)
        return n, (self.filter_images() if filter_ims else None)

    def filter_images(self):
        rec = self.mpsfm_rec
        before = set(rec.registered_images.keys())
        rec.obs.filter_images(0.1, 10.0, 1.0)
        for imid, image in list(rec.registered_images.items()):
            if image.num_points3D == 0:
                rec.obs.deregister_image(imid)
        return before - set(rec.registered_images.keys())

    def find_invalid_depth_points(self, imids):
        out = []
        for imid in imids:
            image = self.mpsfm_rec.images[imid]
            p2 = image.get_observation_point2D_idxs()
            valid = image.depth.valid_at_kps(image.keypoint_coordinates(p2))
            out.append(set(np.array(image.point3D_ids(p2))[~valid].tolist()))
        return out

    def filter_local_points3D(self, local_bundle, max_err, min_angle):
        risky = set.intersection(*self.find_invalid_depth_points(local_bundle["optim_ids"]))
        pts3d = local_bundle["pts3D"]
        if "constpoints" in local_bundle:
            pts3d = pts3d.union(local_bundle["constpoints"])
        n = self.mpsfm_rec.obs.filter_points3D(max_err, 1.5, risky)
        n += self.mpsfm_rec.obs.filter_points3D(max_err, min_angle, pts3d)
        return n

    # -- bundles: base.py:729-749, 799-826 ----------------------------------------------------------------------
    def find_local_bundle(self, refimid, num_images=None):
        rec = self.mpsfm_rec
        optim = set(rec.find_local_bundle_ids(refimid, num_images or self.local_ba_num_images)) | {refimid}
        out = {"ref_id": refimid, "optim_ids": optim}
        lists = [set(rec.images[i].point3D_ids(rec.images[i].get_observation_point2D_idxs())) for i in optim]
        out["pts3D"] = set(rec.images[refimid].point3D_ids(rec.images[refimid].get_observation_point2D_idxs()))
        out["constpoints"] = set.union(*lists) - out["pts3D"]
        return out

    def find_global_bundle(self):
        rec = self.mpsfm_rec
        return {"optim_ids": {i for i, im in rec.images.items() if im.has_pose}, "pts3D": set(rec.points3D.keys()), "constpoints": set()}

    def find_subset_bundle(self, bundle):
        rec = self.mpsfm_rec
        imids = bundle["optim_ids"]
        optim, seen = set(imids), set()
        for imid in imids:
            seen.update(rec.images[imid].point3D_ids(rec.images[imid].get_observation_point2D_idxs()))
        for reg, image in rec.registered_images.items():
            if reg in imids:
                continue
            if set(image.point3D_ids(image.get_observation_point2D_idxs())) & seen:
                optim.add(reg)
        return {"optim_ids": optim, "pts3D": seen}
