"""CPU tests of the ObservationManager-shaped adaptor (mpsfm_amd/sfm/scene/observations.py): its bookkeeping rules
against a brute-force per-point restatement of COLMAP 3.11's ObservationManager, with the oracle's numerics injected
(the HIP numerics are compared with the oracle's in tests/test_gpu_seam.py)."""

import copy

import numpy as np

from numpy_scene import INVALID_POINT3D, ObservationManager, scene_from_problem
from mpsfm_amd.sfm.scene.observations import HipObservationManager, reprojection_decisions
from mpsfm_amd.synthetic import make_scene
from oracle import cpu_oracle as O

EPS = np.finfo(float).eps


def oracle_numerics(tracks, xyz, device):
    return O.filter_tracks(tracks, xyz)


def make_dirty_scene(seed=0, n_cams=8, n_pts=500):
    """A reconstruction with every kind of defect the filters look for."""
    prob, truth = make_scene(n_cams, n_pts, True, seed=seed, perturb=False)
    sc = scene_from_problem(prob, truth, seed=seed)
    rng = np.random.default_rng(seed + 100)
    pids = np.array(sorted(sc.points3D))
    rng.shuffle(pids)
    for pid in pids[:25]:      # behind some of its cameras
        sc.points3D[int(pid)].xyz[:] = sc.points3D[int(pid)].xyz * -4.0 + rng.normal(0, 3, 3)
    for pid in pids[25:75]:    # off by decimetres: many bad observations
        sc.points3D[int(pid)].xyz += rng.normal(0, 0.3, 3)
    for pid in pids[75:100]:   # pushed far along the first viewing ray: tiny parallax
        el = sc.points3D[int(pid)].track.elements[0]
        im = sc.images[el.image_id]
        C = -im.cam_from_world.rotation.matrix().T @ im.cam_from_world.translation
        sc.points3D[int(pid)].xyz[:] = C + (sc.points3D[int(pid)].xyz - C) * 4000.0
    for im in sc.images.values():  # single gross keypoint errors
        idx = rng.choice(len(im.kps), max(1, len(im.kps) // 20), replace=False)
        im.kps[idx] += rng.uniform(-40, 40, (len(idx), 2))
    return sc


def scene_state(sc):
    pts = {pid: (tuple(np.round(p.xyz, 12)), tuple(sorted((e.image_id, e.point2D_idx) for e in p.track.elements))) for pid, p in sc.points3D.items()}
    kp = {imid: im.kp_point3D.copy() for imid, im in sc.images.items()}
    return pts, kp


def assert_same_state(a, b):
    pa, ka = scene_state(a)
    pb, kb = scene_state(b)
    assert pa == pb
    for imid in ka:
        np.testing.assert_array_equal(ka[imid], kb[imid])
    # internal consistency: every keypoint's point lists that keypoint, and the other way round
    for pid, p in a.points3D.items():
        for e in p.track.elements:
            assert a.images[e.image_id].kp_point3D[e.point2D_idx] == pid
    for imid, im in a.images.items():
        for i in np.flatnonzero(im.kp_point3D != INVALID_POINT3D):
            assert any(e.image_id == imid and e.point2D_idx == i for e in a.points3D[int(im.kp_point3D[i])].track.elements)


class BruteForce:
    """Per-point restatement of COLMAP 3.11 ObservationManager::FilterPoints3D / FilterAllPoints3D /
    FilterObservationsWithNegativeDepth on the stand-in bookkeeping (plain Python + NumPy projections)."""

    def __init__(self, sc):
        self.sc, self.obs = sc, ObservationManager(sc)

    def _cam_point(self, image_id, X):
        cfw = self.sc.images[image_id].cam_from_world
        return cfw.rotation.matrix() @ X + cfw.translation

    def _sq_err(self, image_id, idx, X):
        Xc = self._cam_point(image_id, X)
        if Xc[2] < EPS:
            return np.finfo(float).max
        im = self.sc.images[image_id]
        fx, fy, cx, cy = self.sc.rec.cameras[im.camera_id].params
        return float((fx * Xc[0] / Xc[2] + cx - im.kps[idx][0]) ** 2 + (fy * Xc[1] / Xc[2] + cy - im.kps[idx][1]) ** 2)

    def filter_reproj(self, max_err, ids):
        n = 0
        for pid in ids:
            if pid not in self.sc.points3D:
                continue
            p = self.sc.points3D[pid]
            L = p.track.length()
            if L < 2:
                n += L
                self.obs.delete_point3D(pid)
                continue
            bad = [e for e in p.track.elements if self._sq_err(e.image_id, e.point2D_idx, p.xyz) > max_err * max_err]
            if len(bad) >= L - 1:
                n += L
                self.obs.delete_point3D(pid)
            else:
                n += len(bad)
                for e in bad:
                    self.obs.delete_observation(e.image_id, e.point2D_idx)
        return n

    def filter_angle(self, min_angle, ids):
        n = 0
        for pid in ids:
            if pid not in self.sc.points3D:
                continue
            p = self.sc.points3D[pid]
            C = []
            for e in p.track.elements:
                cfw = self.sc.images[e.image_id].cam_from_world
                C.append(-cfw.rotation.matrix().T @ cfw.translation)
            keep = False
            for i in range(len(C)):
                for j in range(i):
                    b2 = np.sum((C[i] - C[j]) ** 2)
                    r1, r2 = np.sum((p.xyz - C[i]) ** 2), np.sum((p.xyz - C[j]) ** 2)
                    den = 2 * np.sqrt(r1 * r2)
                    ang = 0.0 if den == 0 else abs(np.arccos(np.clip((r1 + r2 - b2) / den, -1, 1)))
                    ang = min(ang, np.pi - ang)
                    keep = keep or ang >= np.deg2rad(min_angle)
            if not keep:
                n += 1
                self.obs.delete_point3D(pid)
        return n

    def filter_points3D(self, max_err, min_angle, ids):
        ids = list(ids)
        return self.filter_reproj(max_err, ids) + self.filter_angle(min_angle, ids)

    def filter_all_points3D(self, max_err, min_angle):
        n = self.filter_reproj(max_err, list(self.sc.points3D))
        return n + self.filter_angle(min_angle, list(self.sc.points3D))

    def filter_observations_with_negative_depth(self):
        n = 0
        for imid, im in self.sc.registered_images.items():
            for i in range(len(im.kps)):
                pid = int(im.kp_point3D[i])
                if pid == INVALID_POINT3D:
                    continue
                if not (self._cam_point(imid, self.sc.points3D[pid].xyz)[2] >= EPS):
                    self.obs.delete_observation(imid, i)
                    n += 1
        return n


def _pair(seed):
    a = make_dirty_scene(seed)
    b = copy.deepcopy(a)
    a.obs = HipObservationManager(a, ObservationManager(a), numerics=oracle_numerics)
    return a, b, BruteForce(b)


def test_reprojection_decisions_rules():
    start = np.array([0, 1, 3, 6, 10])
    err = np.array([0.0, 1.0, 99.0, 1.0, 99.0, 99.0, 1.0, 1.0, 99.0, 1.0])
    front = np.array([1, 1, 1, 1, 1, 1, 1, 0, 1, 1], bool)
    whole, bad = reprojection_decisions(start, err, front, 4.0)
    np.testing.assert_array_equal(whole, [True, True, True, False])  # length 1; 1 of 2 bad; 2 of 3 bad; 2 of 4 bad
    np.testing.assert_array_equal(bad, [0, 0, 1, 0, 1, 1, 0, 1, 1, 0])


def test_filter_points3D_equals_bruteforce_colmap_rules():
    a, b, bf = _pair(3)
    ids = sorted(a.points3D)[::2]
    na = a.obs.filter_points3D(4.0, 1.5, set(ids))
    nb = bf.filter_points3D(4.0, 1.5, ids)
    assert na == nb and na > 30
    assert_same_state(a, b)
    assert 0 < len(a.points3D) < 500


def test_filter_all_and_negative_depth_equal_bruteforce():
    a, b, bf = _pair(4)
    na, nb = a.obs.filter_observations_with_negative_depth(), bf.filter_observations_with_negative_depth()
    assert na == nb and na > 5
    assert_same_state(a, b)
    na, nb = a.obs.filter_all_points3D(4.0, 0.001), bf.filter_all_points3D(4.0, 0.001)
    assert na == nb and na > 20
    assert_same_state(a, b)
    # a second pass finds nothing new except what the first pass uncovered (idempotent once stable)
    while a.obs.filter_all_points3D(4.0, 0.001) > 0:
        pass
    assert a.obs.filter_all_points3D(4.0, 0.001) == 0 and a.obs.filter_observations_with_negative_depth() == 0


def test_small_angle_mask_and_passthrough():
    a, b, bf = _pair(5)
    ids = sorted(a.points3D)
    mask = a.obs.find_small_angle_points_mask(1.5, ids)
    assert mask.shape == (len(ids),) and 10 < mask.sum() < len(ids)
    np.testing.assert_array_equal(a.find_points3D_with_small_triangulation_angle(1.5, ids), mask)
    assert a.obs.find_small_angle_points_mask(1.5, []).shape == (0,)
    # bookkeeping calls reach the wrapped manager
    pid = ids[0]
    a.obs.delete_point3D(pid)
    assert pid not in a.points3D
