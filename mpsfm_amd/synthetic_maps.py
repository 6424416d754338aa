"""Synthetic per-image depth / normal priors for the depth-from-normals integration path.

A smooth log-depth surface with a step edge; normals consistent with it under the perspective model
the reference's integration uses (nz_u * dz/du + nx = 0, reference
mpsfm/sfm/scene/image/integration.py:318-325,631-680), stored in the reference's channel convention
(nx = normal[...,1], ny = normal[...,0], nz = -normal[...,2], integration.py:273-275); a noisy,
mis-scaled prior depth with its variance; a few sparse 3-D point depths with variances.
"""

from __future__ import annotations

import numpy as np


def make_maps(H=48, W=64, seed=0, n_sparse=60, prior_noise=0.05, f=None):
    rng = np.random.default_rng(seed)
    f = float(f if f is not None else 1.1 * W)
    cu, cv = (H - 1) / 2.0, (W - 1) / 2.0        # principal point in (vertical, horizontal) map pixels
    rows, cols = np.mgrid[0:H, 0:W].astype(np.float64)
    u = (H - 1 - rows) - cu                          # vertical coordinate, up positive (integration.py:322-324)
    v = cols - cv
    z = (np.log(4.0) + 0.15 * np.sin(2 * np.pi * v / W) * np.cos(2 * np.pi * u / H) + 0.002 * v - 0.001 * u
         + 0.25 * (v > 0.2 * W))                      # log depth with a step edge
    depth = np.exp(z)
    # analytic-free gradients of z along +u (row-1) and +v (col+1), central where possible
    p = np.zeros_like(z); q = np.zeros_like(z)
    p[1:-1] = (z[:-2] - z[2:]) / 2; p[0] = z[0] - z[1]; p[-1] = z[-2] - z[-1]
    q[:, 1:-1] = (z[:, 2:] - z[:, :-2]) / 2; q[:, 0] = z[:, 1] - z[:, 0]; q[:, -1] = z[:, -1] - z[:, -2]
    n = np.stack([-p, -q, (1 + u * p + v * q) / f], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    n += rng.normal(0, 0.01, n.shape)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    normals = np.stack([n[..., 1], n[..., 0], -n[..., 2]], -1)      # reference channel order
    ncov = np.zeros((H, W, 3, 3))
    ncov[..., 0, 0] = ncov[..., 1, 1] = ncov[..., 2, 2] = 0.01**2 * rng.uniform(0.5, 2.0, (H, W))
    prior = 1.2 * depth * np.exp(rng.normal(0, prior_noise, (H, W)))  # scale-wrong, noisy prior
    uncertainty = (prior_noise * prior) ** 2
    valid = rng.uniform(size=(H, W)) > 0.03
    ys, xs = rng.integers(1, H - 1, n_sparse), rng.integers(1, W - 1, n_sparse)
    kps = np.stack([xs, ys], 1)
    depth3d = depth[ys, xs] * np.exp(rng.normal(0, 0.01, n_sparse))
    zvars3d = (0.01 * depth3d) ** 2 * rng.uniform(0.5, 2.0, n_sparse)
    K = (f, f, cu, cv)
    return dict(depth_true=depth, depth_prior=prior, depth_uncertainty=uncertainty, valid=valid, normals=normals,
                normals_uncertainty=ncov, depth_init=prior.copy(), K=K, kps=kps, depth3d=depth3d, zvars3d=zvars3d)
