"""Sampling of per-image prior maps at keypoints.

``bilinear_at_kps`` restates ``PriorUtils._data_at_kps`` (reference
mpsfm/sfm/scene/image/mixins/priorutils.py:49-62): torch ``grid_sample`` with
``mode="bilinear", padding_mode="zeros", align_corners=True`` on a float64 copy of the map, with
keypoints scaled by ``camera.sx, camera.sy``.  With align_corners=True the normalised grid maps
back to pixel coordinates x*sx, y*sy; taps outside the map contribute zero.
"""

from __future__ import annotations

import numpy as np


def bilinear_at_kps(data: np.ndarray, kps: np.ndarray, sx: float, sy: float) -> np.ndarray:
    data = np.asarray(data, dtype=np.float64)
    kps = np.asarray(kps, dtype=np.float64)
    if kps.ndim == 1:
        kps = kps[None]
    H, W = data.shape[:2]
    # grid_sample(align_corners=True): pixel = ((g + 1) / 2) * (size - 1), g = p / (size - 1) * 2 - 1
    x = ((kps[:, 0] * sx) / (W - 1) * 2 - 1 + 1) * 0.5 * (W - 1)
    y = ((kps[:, 1] * sy) / (H - 1) * 2 - 1 + 1) * 0.5 * (H - 1)
    x0, y0 = np.floor(x), np.floor(y)
    wx1, wy1 = x - x0, y - y0
    wx0, wy0 = 1.0 - wx1, 1.0 - wy1
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    out = np.zeros(kps.shape[0])
    for dx, dy, w in ((0, 0, wx0 * wy0), (1, 0, wx1 * wy0), (0, 1, wx0 * wy1), (1, 1, wx1 * wy1)):
        xi, yi = x0 + dx, y0 + dy
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        out[ok] += w[ok] * data[yi[ok], xi[ok]]
    return out


class PriorUtils:
    """Mixin with the reference's accessor names (priorutils.py:21-47)."""

    data = None
    data_prior = None
    uncertainty = None
    valid = None
    camera = None

    def data_prior_at_kps(self, kps):
        return self._data_at_kps(kps, self.data_prior)

    def data_at_kps(self, kps):
        return self._data_at_kps(kps, self.data)

    def uncertainty_at_kps(self, kps):
        return self._data_at_kps(kps, self.uncertainty)

    def valid_at_kps(self, kps):
        return self._data_at_kps(kps, self.valid) == 1

    def _data_at_kps(self, kps, data, mode="bilinear"):
        assert mode == "bilinear"
        return bilinear_at_kps(data, kps, self.camera.sx, self.camera.sy)


def fit_robust_gaussian_mad(data):
    """Median and 1.4826 * MAD (reference bundle_adjustment.py:10-15)."""
    data = np.asarray(data, dtype=np.float64)
    mu = np.median(data)
    return mu, 1.4826 * np.median(np.abs(data - mu))
