"""``Integration`` mixin with the reference's method names (mpsfm/sfm/scene/image/integration.py:80-137,
383-520): ``integrate()`` gathers the image's prior maps and the sparse 3-D points exactly as
``_prepare_integration_variables`` does and runs the IRLS / preconditioned-CG solve on the GPU through
``mpsfm_integrate_depth``; the state the reference caches between calls (integrated, energy_old, wu, wv —
IntVars :18-29) lives on the object.  No CPU fallback."""

from __future__ import annotations

import numpy as np


class Integration:
    conf_integration = dict(
        large_number=1e6, max_iter=10, tol=5e-2, step_size=1, cg_max_iter=5000, cg_tol=1e-3, lambda1=1, lambda2=1, k=1,
        depth_magnitude_multiplier=1, normals_magnitude_multiplier=1, scale_filter=True, scale_filter_factor=1.5,
        robust_triangles=2, downscale_factor=2, downscaled=True, ignore_depths=True, int_cov_rtol=1e-10, int_cov_max_iter=50000,
    )

    def __init__(self):
        self.integrated = False
        self.energy_old = None
        self.wu = None
        self.wv = None
        self.count_integrated = 0
        self.count_skipped = 0
        self.last_integration_summary = None
        self.Hessian = None

    # the host object provides: self.mpsfm_rec, self.imid, self.image, self.camera (with sx, sy,
    # calibration_matrix()), self.depth (data, data_prior, uncertainty, valid), self.normals (data, uncertainty)
    def _prepare_integration_variables(self):
        """reference :90-131"""
        conf = self.conf_integration
        _, pts3dids, kps, depth3d, success = self.mpsfm_rec.project_image_3d_points(self.imid)
        if not success:
            return None, False
        pts3dids = np.array(pts3dids)
        if conf["robust_triangles"] is not None and len(pts3dids):
            safe = ~self.mpsfm_rec.find_points3D_with_small_triangulation_angle(min_angle=conf["robust_triangles"], point3D_ids=pts3dids)
            pts3dids, kps, depth3d = pts3dids[safe], kps[safe], depth3d[safe]
        kps = kps * np.array([self.camera.sx, self.camera.sy])
        kps = (kps + 0.5).astype(int)
        if len(pts3dids) == 0:
            zvars3d, mask = np.array([]), slice(None)
        else:
            _, zvars3d = self.mpsfm_rec.point_covs.points_zvars(self.image, list(pts3dids))
            x, y = kps.T
            mask = (x >= 0) & (x < self.depth.data.shape[1]) & (y >= 0) & (y < self.depth.data.shape[0])
        Kc = self.camera.calibration_matrix()
        return dict(kps=kps[mask], zvars3d=np.asarray(zvars3d)[mask], depth3d=np.asarray(depth3d)[mask],
                    K=[Kc[1, 1] * self.camera.sy, Kc[0, 0] * self.camera.sx, Kc[1, 2] * self.camera.sy, Kc[0, 2] * self.camera.sx]), True

    def _solver_conf(self):
        skip = ("robust_triangles", "downscale_factor", "downscaled", "ignore_depths", "int_cov_rtol", "int_cov_max_iter")
        return {k: v for k, v in self.conf_integration.items() if k not in skip}

    def integrate(self, cache_device="cpu"):
        """Integrate depth map from normals with depth constraints (reference :133-137)."""
        assert self.image.has_pose and self.depth.activated, "Image not registered or depth map not activated"
        kwargs, _ = self._prepare_integration_variables()
        return self._integrate(cache_device=cache_device, **kwargs)

    def _integrate(self, depth3d, zvars3d, kps, K, cache_device="cpu", init=True):
        from ... import capi

        nunc = np.asarray(self.normals.uncertainty)
        nvar = np.stack([nunc[..., 0, 0], nunc[..., 1, 1], nunc[..., 2, 2]], -1) if nunc.ndim == 4 else nunc
        conf = self._solver_conf()
        depth, summary, wu, wv = capi.integrate_depth(
            self.depth.data_prior, self.depth.uncertainty, self.depth.valid, self.normals.data, nvar, self.depth.data, K, kps,
            depth3d, zvars3d, conf=conf, init=init, integrated=self.integrated, energy_old=self.energy_old or 0.0,
            wu=self.wu, wv=self.wv)
        self.last_integration_summary = summary
        self.integrated, self.energy_old, self.wu, self.wv = summary["integrated"], summary["energy_old"], wu, wv
        if depth is None:
            self.count_integrated += 1
            return False
        self.count_skipped += 1
        self.depth.data = depth
        return True

    # ---- uncertainty propagation (reference :522-629) ----------------------------------------------------
    def calculate_hessian(self, downscaled, ignore_depths=None):
        """reference :522-574.  The matrix itself is never assembled here (the device solve is matrix-free):
        `self.Hessian` holds the inputs that define it plus, once computed, the field H^-1 1 from which every
        query of `calculate_int_covs_at_points` is a gather."""
        conf = self.conf_integration
        if ignore_depths is None:
            ignore_depths = conf["ignore_depths"]
        kwargs, _ = self._prepare_integration_variables()
        depth3d, zvars3d, kps, K = kwargs["depth3d"], kwargs["zvars3d"], kwargs["kps"], list(kwargs["K"])
        prior, unc, valid, ckpt = self.depth.data_prior, self.depth.uncertainty, self.depth.valid, self.depth.data
        normals, nunc = self.normals.data, np.asarray(self.normals.uncertainty)
        full_shape = np.asarray(self.depth.data).shape
        if downscaled:
            fac = conf["downscale_factor"]
            H, W = np.asarray(prior).shape
            size = (int(W // fac), int(H // fac))
            kps = (kps // fac).astype(int)
            K = [v / fac for v in K]
            prior, unc, ckpt = resize_linear(prior, size), resize_linear(unc, size), resize_linear(ckpt, size)
            valid = np.floor(resize_linear(np.asarray(valid, np.float64), size) + 0.5) != 0   # uint8 resize, then astype(bool)
            normals, nunc = self.normals.data_downscaled, np.asarray(self.normals.uncertainty_downscaled)
        if not ignore_depths and len(kps):
            # :283 ravels the (possibly downscaled) pixels with the FULL map shape; the flat ids index the solved map
            ids = np.ravel_multi_index((kps[:, 1], kps[:, 0]), full_shape)
            Hs, Ws = np.asarray(prior).shape
            if ids.max() >= Hs * Ws:
                raise IndexError("sparse id out of bounds for the downscaled map (same failure as the reference)")
            kps = np.stack([ids % Ws, ids // Ws], 1)
        nvar = np.stack([nunc[..., 0, 0], nunc[..., 1, 1], nunc[..., 2, 2]], -1) if nunc.ndim == 4 else nunc
        self.Hessian = dict(depth_prior=prior, depth_uncertainty=unc, valid=valid, normals=normals, normals_var=nvar,
                            depth_checkpoint=ckpt, K=K, kps=kps, depth3d=depth3d, zvars3d=zvars3d,
                            use_sparse=not ignore_depths, shape=np.asarray(prior).shape, field=None, summary=None)
        return self.Hessian

    def calculate_int_covs_at_points(self, pts, verbose=False, Hessian=None, downscaled=None, ignore_depths=None):
        """reference :576-600 + IntegrationUncertainty.solve :62-78 (sum of the solution column per query)."""
        from ... import capi

        conf = self.conf_integration
        if downscaled is None:
            downscaled = conf["downscaled"]
        if Hessian is None:
            if self.Hessian is None:
                self.calculate_hessian(downscaled=downscaled, ignore_depths=ignore_depths)
            Hessian = self.Hessian
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        kps = pts // conf["downscale_factor"] if downscaled else pts
        xy = np.round(kps).astype(int)
        Hs, Ws = Hessian["shape"]
        np.ravel_multi_index(xy.T[::-1], (Hs, Ws))  # raises ValueError on an out-of-map query like the reference
        if Hessian["field"] is None:
            h = Hessian
            _, summary, field = capi.integration_variances(
                h["depth_prior"], h["depth_uncertainty"], h["valid"], h["normals"], h["normals_var"], h["depth_checkpoint"], h["K"],
                np.zeros((0, 2), int), kps=h["kps"], depth3d=h["depth3d"], zvars3d=h["zvars3d"], use_sparse=h["use_sparse"],
                conf=self._solver_conf(), rtol=conf["int_cov_rtol"], max_iter=conf["int_cov_max_iter"], return_field=True)
            h["field"], h["summary"] = field, summary
        return Hessian["field"][xy[:, 1], xy[:, 0]]

    def calculate_int_covs_at_kps(self, Hessian=None, pts2d=None, downscaled=None):
        """reference :602-616"""
        kps = self.mpsfm_rec.keypoints(self.imid)
        if pts2d is None:
            pts2d = np.arange(len(kps))
        else:
            kps = kps[pts2d]
        kps_down = kps * np.array([self.camera.sx, self.camera.sy])
        log_uncert = self.calculate_int_covs_at_points(kps_down, Hessian=Hessian, downscaled=downscaled)
        uncert = log_uncert * self.depth.data_prior_at_kps(kps) ** 2  # var(log d) = var(d) / d^2
        self.depth.uncertainty_update[pts2d] = uncert
        return uncert

    def calculate_int_covs_for_entire_image(self, downscaled=None, ignore_depths=False):
        """reference :618-629"""
        nshape = np.asarray(self.depth.data).shape
        xx, yy = np.meshgrid(np.arange(nshape[1]), np.arange(nshape[0]))
        v = self.calculate_int_covs_at_points(np.array([xx.flatten(), yy.flatten()]).T, downscaled=downscaled,
                                              ignore_depths=ignore_depths)
        return v.reshape(nshape) * np.asarray(self.depth.data) ** 2


def _linear_taps(n_src, n_dst):
    """[n_dst, n_src] interpolation matrix of OpenCV's INTER_LINEAR along one axis: sample position
    (i + 0.5) * n_src / n_dst - 0.5, replicated border, float32 coefficients."""
    pos = (np.arange(n_dst) + 0.5) * (1.0 / (n_dst / n_src)) - 0.5
    lo = np.floor(pos).astype(int)
    frac = (pos - lo).astype(np.float32).astype(np.float64)
    frac[(lo < 0) | (lo >= n_src - 1)] = 0.0
    lo = np.clip(lo, 0, n_src - 1)
    T = np.zeros((n_dst, n_src))
    T[np.arange(n_dst), lo] += 1.0 - frac
    T[np.arange(n_dst), np.minimum(lo + 1, n_src - 1)] += frac
    return T


def resize_linear(img, dsize):
    """`cv2.resize(img, (w, h))` (bilinear) for one float64 channel — what the reference applies to the depth
    prior, its variance and the checkpoint when `downscaled` (:241-259, :293-297).  cv2 is not available
    offline: restated from OpenCV's sampling rule, exact 2x2 means for even sizes and factor 2."""
    img = np.asarray(img, dtype=np.float64)
    return _linear_taps(img.shape[0], int(dsize[1])) @ img @ _linear_taps(img.shape[1], int(dsize[0])).T



def integrate_bundle(images, workers: int = 8, cache_device="cpu", batched: bool = True):
    """Integrate several images (what MpsfmMapper.integrate_bundle loops over, reference
    mpsfm/sfm/mapper/base.py:619-631).  Images whose maps have the same size and configuration go through ONE
    mpsfm_integrate_depth_batch call: every image keeps its own IRLS / CG state on the device, so the results
    equal integrating one image after the other while launch and synchronisation latency is paid once.
    `batched=False` falls back to concurrent single-image calls on `workers` host threads (each on a HIP
    stream of its own).  Returns the per-image `changed` flags."""
    from concurrent.futures import ThreadPoolExecutor

    from ... import capi

    images = list(images)
    if not batched:
        if workers <= 1 or len(images) <= 1:
            return [im.integrate(cache_device=cache_device) for im in images]
        with ThreadPoolExecutor(max_workers=min(workers, len(images))) as ex:
            return list(ex.map(lambda im: im.integrate(cache_device=cache_device), images))
    groups: dict = {}
    for idx, im in enumerate(images):
        assert im.image.has_pose and im.depth.activated, "Image not registered or depth map not activated"
        kwargs, _ = im._prepare_integration_variables()
        nunc = np.asarray(im.normals.uncertainty)
        nvar = np.stack([nunc[..., 0, 0], nunc[..., 1, 1], nunc[..., 2, 2]], -1) if nunc.ndim == 4 else nunc
        item = dict(depth_prior=im.depth.data_prior, depth_uncertainty=im.depth.uncertainty, valid=im.depth.valid, normals=im.normals.data,
                    normals_var=nvar, depth_init=im.depth.data, K=kwargs["K"], kps=kwargs["kps"], depth3d=kwargs["depth3d"],
                    zvars3d=kwargs["zvars3d"], init=True, integrated=im.integrated, energy_old=im.energy_old or 0.0, wu=im.wu, wv=im.wv)
        conf = im._solver_conf()
        key = (np.asarray(im.depth.data).shape, tuple(sorted(conf.items())))
        groups.setdefault(key, []).append((idx, im, item, conf))
    changed = [False] * len(images)
    for members in groups.values():
        res = capi.integrate_depth_batch([m[2] for m in members], conf=members[0][3])
        for (idx, im, _, _), (depth, summary, wu, wv) in zip(members, res):
            im.last_integration_summary = summary
            im.integrated, im.energy_old, im.wu, im.wv = summary["integrated"], summary["energy_old"], wu, wv
            if depth is None:
                im.count_integrated += 1
                continue
            im.count_skipped += 1
            im.depth.data = depth
            changed[idx] = True
    return changed
