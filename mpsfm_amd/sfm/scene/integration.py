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
        robust_triangles=2,
    )

    def __init__(self):
        self.integrated = False
        self.energy_old = None
        self.wu = None
        self.wv = None
        self.count_integrated = 0
        self.count_skipped = 0
        self.last_integration_summary = None

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

    def integrate(self, cache_device="cpu"):
        """Integrate depth map from normals with depth constraints (reference :133-137)."""
        assert self.image.has_pose and self.depth.activated, "Image not registered or depth map not activated"
        kwargs, _ = self._prepare_integration_variables()
        return self._integrate(cache_device=cache_device, **kwargs)

    def _integrate(self, depth3d, zvars3d, kps, K, cache_device="cpu", init=True):
        from ... import capi

        nunc = np.asarray(self.normals.uncertainty)
        nvar = np.stack([nunc[..., 0, 0], nunc[..., 1, 1], nunc[..., 2, 2]], -1) if nunc.ndim == 4 else nunc
        conf = {k: v for k, v in self.conf_integration.items() if k != "robust_triangles"}
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


def integrate_bundle(images, workers: int = 8, cache_device="cpu"):
    """Integrate several images concurrently (what MpsfmMapper.integrate_bundle loops over, reference
    mpsfm/sfm/mapper/base.py:619-631).  Every call of mpsfm_integrate_depth runs on a HIP stream of its
    own and ctypes releases the GIL, so the per-image launch sequences overlap on the GPU; results are
    identical to integrating one image after the other.  Returns the per-image `changed` flags."""
    from concurrent.futures import ThreadPoolExecutor

    images = list(images)
    if workers <= 1 or len(images) <= 1:
        return [im.integrate(cache_device=cache_device) for im in images]
    with ThreadPoolExecutor(max_workers=min(workers, len(images))) as ex:
        return list(ex.map(lambda im: im.integrate(cache_device=cache_device), images))
