"""Small geometry helpers with the reference's names (mpsfm/utils/geometry.py:6-19, 54-75)."""

from __future__ import annotations

import numpy as np


def project3D(points3D, H, K):
    """World points -> (pixels, depth) with a 4x4 cam_from_world H and calibration K."""
    points3D = np.asarray(points3D, dtype=np.float64)
    cam = points3D @ H[:3, :3].T + H[:3, 3]
    depth = cam[:, 2].copy()
    return ((cam / depth[:, None]) @ K.T)[:, :2], depth


def project3D_colmap(image, camera, points3D):
    H = np.concatenate([image.cam_from_world.matrix(), np.array([[0, 0, 0, 1.0]])], axis=0)
    return project3D(points3D, H, camera.calibration_matrix())


def calculate_triangulation_angle(proj_center1, proj_center2, point3D):
    """min(angle, pi - angle) between the two viewing rays (law of cosines on squared lengths)."""
    b2 = np.sum((proj_center1 - proj_center2) ** 2)
    r1 = np.sum((point3D - proj_center1) ** 2)
    r2 = np.sum((point3D - proj_center2) ** 2)
    den = 2.0 * np.sqrt(r1 * r2)
    if den == 0.0:
        return 0.0
    ang = np.abs(np.arccos(np.clip((r1 + r2 - b2) / den, -1.0, 1.0)))
    return min(ang, np.pi - ang)


def has_point_positive_depth(cam_from_world, point3D, return_depth=False):
    depth = float(np.dot(cam_from_world[2, :], np.append(point3D, 1.0)))
    ok = depth >= np.finfo(float).eps
    return (ok, depth) if return_depth else ok
