"""NumPy restatement of the depth-block selection arithmetic — TEST INFRASTRUCTURE, not product code.

Follows reference mpsfm/sfm/mapper/bundle_adjustment.py:130-161 (masks and loss weights of the depth blocks) and
:312-329 (whitened log-depth errors of update_truncation_multiplier) per observation, with the map sampling of
PriorUtils._data_at_kps (image/mixins/priorutils.py:49-62) through mpsfm_amd.sfm.scene.priorutils.bilinear_at_kps —
which tests/test_reference_fixtures_cpu.py pins against vectors computed by the reference's own PriorUtils — and the
camera-frame depth of geometry.project3D (pinned the same way).  Checker of the HIP kernel k_depth_blocks.
"""

from __future__ import annotations

import numpy as np

from mpsfm_amd.sfm.scene.priorutils import bilinear_at_kps
from mpsfm_amd.synthetic import R_from_quat

F_VALID, F_POSITIVE, F_SCALE, F_GROSS = 1, 2, 4, 8


def depth_blocks(depth_maps, valid_maps, sx, sy, cam_quat, cam_t, obs_img, obs_xy, obs_var, obs_pt, pts, scale_filter_factor=1.5,
                       multiplier=2.0):
    """Same arguments and outputs as mpsfm_amd.capi.depth_blocks."""
    n = len(obs_img)
    out = {k: np.zeros(n) for k in ("depth", "depth3d", "magnitude", "param", "whitened")}
    flags = np.zeros(n, np.uint8)
    R = R_from_quat(np.asarray(cam_quat, np.float64).reshape(-1, 4))
    obs_xy = np.asarray(obs_xy, np.float64).reshape(-1, 2)
    pts = np.asarray(pts, np.float64).reshape(-1, 3)
    for k in range(len(depth_maps)):
        sel = np.flatnonzero(np.asarray(obs_img) == k)
        if len(sel) == 0:
            continue
        v = bilinear_at_kps(np.asarray(valid_maps[k]), obs_xy[sel], sx[k], sy[k])
        d = bilinear_at_kps(depth_maps[k], obs_xy[sel], sx[k], sy[k])
        X = pts[np.asarray(obs_pt)[sel]]
        z = (np.concatenate([R[k], np.asarray(cam_t, np.float64).reshape(-1, 3)[k][:, None]], 1) @ np.concatenate([X, np.ones((len(X), 1))], 1).T).T[:, 2]
        var = np.asarray(obs_var, np.float64)[sel]
        with np.errstate(divide="ignore", invalid="ignore"):
            div = d / z
            f = (v == 1).astype(np.uint8) * F_VALID | (d > 0).astype(np.uint8) * F_POSITIVE
            f |= ((div < scale_filter_factor) & (div > 1 / scale_filter_factor)).astype(np.uint8) * F_SCALE
            wh_g = np.abs(np.log(d).clip(1e-6, None) - np.log(z).clip(1e-6, None)) / var**0.5
            f |= (wh_g < 3).astype(np.uint8) * F_GROSS
            out["magnitude"][sel] = d**2 * (1 / var.clip(1e-6, None))
            out["param"][sel] = multiplier * var**0.5 / d
            out["whitened"][sel] = (np.log(d) - np.log(z)) / np.clip(var**0.5 / d, 1e-6, None)
        flags[sel] = f
        out["depth"][sel], out["depth3d"][sel] = d, z
    out["flags"] = flags
    return out
