"""Geometry helpers with the reference's names and results (mpsfm/utils/geometry.py:6-75).

Every function is pinned by tests/golden/reference_geometry_pointcov.npz, which was produced by importing the
reference's own file (tests/golden/make_golden_reference.py).  Quirks of the reference are kept on purpose:
``calculate_triangulation_angle`` feeds the law of cosines with lengths where COLMAP's
CalculateTriangulationAngle uses squared lengths (reference :54-65) — the HIP kernels (mpsfm_filter_tracks)
follow COLMAP, as the fork's C++ ObservationManager does; this Python helper follows the Python reference.
"""

from __future__ import annotations

import numpy as np


def project3D(points3D, H, K):
    """World points [N,3] -> (pixels [N,2], depth [N]) for a 4x4 (or 3x4) cam_from_world H and calibration K."""
    X = np.asarray(points3D, dtype=np.float64)
    H = np.asarray(H, dtype=np.float64)
    cam = (H[:3, :] @ np.concatenate([X, np.ones((X.shape[0], 1))], axis=1).T).T
    depth = cam[:, 2].copy()
    return (K @ (cam / depth[:, None]).T).T[:, :2], depth


def project3D_colmap(image, camera, points3D):
    """reference :6-10 — pose and calibration taken from the (pycolmap-shaped) image / camera objects."""
    H = np.vstack([image.cam_from_world.matrix(), [0.0, 0.0, 0.0, 1.0]])
    return project3D(points3D, H, camera.calibration_matrix())


def unproject_to_cam(xy_depth, K):
    """[3,N] rows (x d, y d, d) -> homogeneous camera points [N,4] (reference :47-51)."""
    p = np.linalg.inv(K) @ xy_depth
    return np.vstack([p, np.ones((1, p.shape[1]))]).T


def unproject_to_world(xy_depth, K, H):
    """reference :40-44; H is world_from_cam (4x4)."""
    return (H @ unproject_to_cam(xy_depth, K).T).T[:, :3]


def unproject_depth_map_to_world(depth, K, H, mask=None):
    """Every (masked) pixel of a depth map lifted to world coordinates, row-major pixel order (reference :22-37)."""
    h, w = depth.shape
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    x, y, d = x.ravel(), y.ravel(), depth.ravel()
    if mask is not None:
        m = mask.ravel()
        x, y, d = x[m], y[m], d[m]
    return unproject_to_world(np.vstack([x * d, y * d, d]), K, H)


def calculate_triangulation_angle(proj_center1, proj_center2, point3D):
    """reference :54-65, including its use of (unsquared) norms in the cosine rule."""
    base = np.linalg.norm(proj_center1 - proj_center2)
    ray1 = np.linalg.norm(point3D - proj_center1)
    ray2 = np.linalg.norm(point3D - proj_center2)
    den = 2.0 * np.sqrt(ray1 * ray2)
    if den == 0.0:
        return 0.0
    ang = np.abs(np.arccos((ray1 + ray2 - base) / den))
    return min(ang, np.pi - ang)


def has_point_positive_depth(cam_from_world, point3D, return_depth=False):
    """depth >= eps with depth = third row of the 3x4 pose times the homogeneous point (reference :68-75)."""
    depth = np.dot(cam_from_world[2, :], np.append(point3D, 1))
    ok = depth >= np.finfo(float).eps
    return (ok, depth) if return_depth else ok
