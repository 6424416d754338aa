"""A NumPy scene that exposes the accessor names the reference's Optimizer / MpsfmTriangulator use on
``MpsfmReconstruction`` and the fork's pycolmap objects (SURVEY.md §8b "Scene accessors"):

  mpsfm_rec.images[imid]            .camera_id .kp_std .cam_from_world(.rotation.quat xyzw, .translation)
                                    .depth .has_pose .points2D[i].xy
                                    .get_observation_point2D_idxs() .keypoint_coordinates(idxs)
                                    .point3D_ids(idxs)
  mpsfm_rec.rec.cameras[camera_id]  .params [fx fy cx cy] .sx .sy .cam_from_img(xy) .calibration_matrix()
  mpsfm_rec.points3D[pid]           .xyz .track.length() .track.elements[*].image_id/.point2D_idx
  mpsfm_rec.obs                     .add_point3D(xyz, track) .delete_point3D(pid)
                                    .find_small_angle_points_mask(min_angle_deg, ids)
  mpsfm_rec.point_covs.data         {pid: 3x3}
  mpsfm_rec.project_image_3d_points(imid, ids) / point3D_coordinates(ids) / registered_images

It lets the host layer be exercised without pycolmap (not installable offline); with pycolmap
present the same Optimizer works on the real objects because it only uses these names.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from mpsfm_amd.synthetic import R_from_quat, quat_from_R
from mpsfm_amd.utils.geometry import project3D_colmap
from mpsfm_amd.sfm.scene.observations import HipObservationManager
from mpsfm_amd.sfm.scene.priorutils import PriorUtils

INVALID_POINT3D = 18446744073709551615  # pycolmap's kInvalidPoint3DId (reference triangulator.py:109)


class Rotation3d:
    def __init__(self, quat_xyzw):
        self.quat = np.array(quat_xyzw, dtype=np.float64)

    def matrix(self):
        return R_from_quat(self.quat)[0]


class Rigid3d:
    """cam_from_world: x_cam = R x_world + t."""

    def __init__(self, quat_xyzw, translation):
        self.rotation = Rotation3d(quat_xyzw)
        self.translation = np.array(translation, dtype=np.float64)

    def matrix(self):
        return np.concatenate([self.rotation.matrix(), self.translation[:, None]], axis=1)

    def inverse(self):
        R = self.rotation.matrix()
        return Rigid3d(quat_from_R(R.T[None])[0], -R.T @ self.translation)

    def __mul__(self, pts):
        pts = np.asarray(pts, dtype=np.float64)
        return pts @ self.rotation.matrix().T + self.translation


class NumpyCamera:
    def __init__(self, camera_id, params, width, height, map_width, map_height):
        self.camera_id = camera_id
        self.params = np.array(params, dtype=np.float64)  # PINHOLE fx fy cx cy
        self.width, self.height = width, height
        self.sx, self.sy = map_width / width, map_height / height

    focal_length_x = property(lambda s: s.params[0])
    focal_length_y = property(lambda s: s.params[1])
    principal_point_x = property(lambda s: s.params[2])
    principal_point_y = property(lambda s: s.params[3])

    def calibration_matrix(self):
        fx, fy, cx, cy = self.params
        return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])

    def cam_from_img(self, xy):
        xy = np.atleast_2d(np.asarray(xy, dtype=np.float64))
        return (xy - self.params[2:4]) / self.params[0:2]


@dataclass
class TrackElement:
    image_id: int
    point2D_idx: int


class Track:
    def __init__(self):
        self.elements: list[TrackElement] = []

    def add_element(self, image_id, point2D_idx):
        self.elements.append(TrackElement(int(image_id), int(point2D_idx)))

    def length(self):
        return len(self.elements)


class NumpyPoint3D:
    """A 3-D point whose coordinates live in the scene's columnar store (bulk reads / writes of many points are one
    NumPy gather); ``xyz`` is a writable view of its row, like pycolmap's Point3D.xyz is a view of the C++ object."""

    def __init__(self, xyz, track=None, store=None, row=-1):
        self._store, self._row = store, row
        if store is None:
            self._own = np.array(xyz, dtype=np.float64)
        else:
            store.xyz[row] = xyz
        self.error = -1.0
        self.track = track if track is not None else Track()

    @property
    def xyz(self):
        return self._own if self._store is None else self._store.xyz[self._row]

    @xyz.setter
    def xyz(self, value):
        self.xyz[:] = value


class _PointStore:
    """Rows of xyz / track lengths indexed by point id through a lookup table (ids are small integers here)."""

    def __init__(self, capacity=1 << 16):
        self.xyz = np.zeros((capacity, 3))
        self.tlen = np.zeros(capacity, np.int64)
        self.row_of = np.full(capacity, -1, np.int64)  # id -> row
        self.free = []
        self.n = 0

    def _grow(self, need_rows, need_ids):
        if need_rows > len(self.xyz):
            cap = max(need_rows, 2 * len(self.xyz))
            # views handed out before a growth keep pointing at the old block: growth happens only when points are added
            self.xyz = np.concatenate([self.xyz, np.zeros((cap - len(self.xyz), 3))])
            self.tlen = np.concatenate([self.tlen, np.zeros(cap - len(self.tlen), np.int64)])
        if need_ids > len(self.row_of):
            self.row_of = np.concatenate([self.row_of, np.full(max(need_ids, 2 * len(self.row_of)) - len(self.row_of), -1, np.int64)])

    def add(self, pid):
        row = self.free.pop() if self.free else self.n
        if row == self.n:
            self.n += 1
        self._grow(self.n, pid + 1)
        self.row_of[pid] = row
        return row

    def remove(self, pid):
        self.free.append(int(self.row_of[pid]))
        self.row_of[pid] = -1

    def rows(self, ids):
        return self.row_of[np.asarray(ids, dtype=np.int64)]


class Point2D:
    __slots__ = ("_img", "_i")

    def __init__(self, img, i):
        self._img, self._i = img, i

    @property
    def xy(self):
        return self._img.kps[self._i]

    @property
    def point3D_id(self):
        return int(self._img.kp_point3D[self._i])

    def has_point3D(self):
        return self.point3D_id != INVALID_POINT3D


class _Points2DView:
    def __init__(self, img):
        self._img = img

    def __getitem__(self, i):
        return Point2D(self._img, int(i))

    def __len__(self):
        return self._img.kps.shape[0]

    def __iter__(self):
        return (Point2D(self._img, i) for i in range(len(self)))


class NumpyDepth(PriorUtils):
    """Prior depth map + validity + variance with the attribute names of the reference's Depth
    (mpsfm/sfm/scene/image/depth.py:34-130)."""

    def __init__(self, data_prior, uncertainty, valid, camera, kps):
        self.data_prior = np.array(data_prior, dtype=np.float64)
        self.data = self.data_prior.copy()  # the integrated ("update") map starts as the prior
        self.uncertainty = np.array(uncertainty, dtype=np.float64)
        self.valid = np.array(valid, dtype=np.float64)
        self.camera = camera
        self.kps = kps
        self.scale = 1.0
        self.shift = 0.0
        self.activated = True
        self.uncertainty_update = self.uncertainty_at_kps(kps)  # depth.py:130

    def reset(self):
        """back to the unscaled, not activated prior (reference depth.py:132-140)"""
        self.data_prior = self.data_prior / self.scale
        self.uncertainty = self.uncertainty / self.scale**2
        self.uncertainty_update = self.uncertainty_at_kps(self.kps)
        self.scale, self.shift, self.activated, self.data = 1.0, 0.0, False, None


class NumpyImage:
    def __init__(self, image_id, camera_id, cam_from_world, kps, kp_std=1.0, depth=None):
        self.image_id, self.camera_id = image_id, camera_id
        self.cam_from_world = cam_from_world
        self.kps = np.array(kps, dtype=np.float64).reshape(-1, 2)
        self.kp_point3D = np.full(self.kps.shape[0], INVALID_POINT3D, dtype=np.uint64)
        self.kp_std = kp_std
        self.depth = depth
        self.has_pose = True
        self.points2D = _Points2DView(self)

    def get_observation_point2D_idxs(self):
        return np.flatnonzero(self.kp_point3D != INVALID_POINT3D)

    @property
    def num_points3D(self):
        return int((self.kp_point3D != INVALID_POINT3D).sum())

    def keypoint_coordinates(self, idxs):
        return self.kps[np.asarray(idxs, dtype=np.int64)]

    def point3D_ids(self, idxs=None):
        """pycolmap returns a list of ints; an index ARRAY gets an id array back (no per-element Python objects)."""
        if idxs is None:
            return [int(v) for v in self.kp_point3D]
        if isinstance(idxs, np.ndarray):
            return self.kp_point3D[idxs.astype(np.int64, copy=False)]
        return [int(v) for v in self.kp_point3D[np.asarray(idxs, dtype=np.int64)]]


class PointCovs:
    """reference mpsfm/sfm/scene/pointcov.py:4-20 (pinned by tests/golden/reference_geometry_pointcov.npz).
    The reference evaluates (R^T C R)[2, 2] with R = cam_from_world rotation — kept as it is."""

    def __init__(self):
        self.data = {}

    def points_zvars(self, image, p3d_ids=None):
        if p3d_ids is None:
            p3d_ids = [p.point3D_id for p in image.points2D if p.has_point3D()]
        R = image.cam_from_world.rotation.matrix()
        data = np.array([self.data[p] for p in p3d_ids])
        return p3d_ids, np.einsum("ji,njk,kl->nil", R, data, R)[:, 2, 2]


class _Rec:
    def __init__(self):
        self.cameras = {}


class ObservationManager:
    def __init__(self, scene):
        self.scene = scene

    def add_point3D(self, xyz, track):
        s = self.scene
        pid = s._next_point3D_id
        s._next_point3D_id += 1
        row = s._store.add(pid)
        s.points3D[pid] = NumpyPoint3D(xyz, track, s._store, row)
        s._store.tlen[row] = track.length()
        for el in track.elements:
            s.images[el.image_id].kp_point3D[el.point2D_idx] = pid
        return pid

    def delete_point3D(self, pid):
        s = self.scene
        for el in s.points3D[pid].track.elements:
            s.images[el.image_id].kp_point3D[el.point2D_idx] = INVALID_POINT3D
        del s.points3D[pid]
        s._store.remove(pid)
        s.point_covs.data.pop(pid, None)

    def add_observation(self, pid, track_el):
        """COLMAP ObservationManager::AddObservation"""
        s = self.scene
        s.points3D[pid].track.add_element(track_el.image_id, track_el.point2D_idx)
        s._store.tlen[s._store.row_of[pid]] = s.points3D[pid].track.length()
        s.images[track_el.image_id].kp_point3D[track_el.point2D_idx] = pid

    def delete_observation(self, image_id, point2D_idx):
        """COLMAP ObservationManager::DeleteObservation: a track of length <= 2 goes away with its point."""
        s = self.scene
        pid = int(s.images[image_id].kp_point3D[point2D_idx])
        track = s.points3D[pid].track
        if track.length() <= 2:
            self.delete_point3D(pid)
            return
        track.elements = [el for el in track.elements if not (el.image_id == image_id and el.point2D_idx == point2D_idx)]
        s._store.tlen[s._store.row_of[pid]] = track.length()
        s.images[image_id].kp_point3D[point2D_idx] = INVALID_POINT3D

    def deregister_image(self, image_id):
        self.scene.images[image_id].has_pose = False

    def filter_images(self, min_focal_length_ratio, max_focal_length_ratio, max_extra_param):
        """COLMAP deregisters images without points or with degenerate intrinsics; the stand-in has fixed PINHOLE
        cameras, so only the first rule applies."""
        n = 0
        for imid, im in self.scene.registered_images.items():
            if len(im.get_observation_point2D_idxs()) == 0:
                self.deregister_image(imid)
                n += 1
        return n


class NumpyCorrespondenceGraph:
    """Stand-in for pycolmap.CorrespondenceGraph with the method names the reference uses on it
    (mpsfm/sfm/scene/correspondences/base.py:55-61, 124-138)."""

    def __init__(self):
        self.num_kp, self.matches = {}, {}

    def add_image(self, image_id, num_points2D):
        self.num_kp[int(image_id)] = int(num_points2D)

    def add_correspondences(self, image_id1, image_id2, matches):
        m = np.asarray(matches, np.int64).reshape(-1, 2)
        a, b = int(image_id1), int(image_id2)
        if a > b:
            a, b, m = b, a, m[:, ::-1]
        self.matches[(a, b)] = np.concatenate([self.matches[(a, b)], m]) if (a, b) in self.matches else m.copy()

    def finalize(self):
        pass

    def image_pairs(self):
        return sorted(self.matches)

    def find_correspondences_between_images(self, image_id1, image_id2):
        a, b = int(image_id1), int(image_id2)
        if a <= b:
            return self.matches.get((a, b), np.zeros((0, 2), np.int64))
        return self.matches.get((b, a), np.zeros((0, 2), np.int64))[:, ::-1]

    def num_correspondences_between_images(self, image_id1, image_id2):
        return len(self.find_correspondences_between_images(image_id1, image_id2))

    def num_correspondences_for_image(self, image_id):
        return sum(len(m) for (a, b), m in self.matches.items() if int(image_id) in (a, b))


class NumpyReconstruction:
    Track, TrackElement = Track, TrackElement  # the element types obs.add_point3D / add_observation take (pycolmap.Track / TrackElement)

    def __init__(self):
        self.images: dict[int, NumpyImage] = {}
        self.points3D: dict[int, NumpyPoint3D] = {}
        self.rec = _Rec()
        # bookkeeping by the stand-in manager, per-track numerics by the HIP kernels (as on the real pycolmap object)
        self.obs = HipObservationManager(self, ObservationManager(self))
        self.point_covs = PointCovs()
        self._next_point3D_id = 1
        self._store = _PointStore()

    @property
    def registered_images(self):
        return {i: im for i, im in self.images.items() if im.has_pose}

    def keypoints(self, imid):
        """keypoints of image (reference reconstruction/base.py:85-87)"""
        return self.images[imid].kps.copy()

    def point3D_coordinates(self, ids):
        """xyz of many points (reference reconstruction/base.py point3D_coordinates): one gather from the store"""
        return self._store.xyz[self._store.rows(ids)].reshape(-1, 3)

    # optional bulk accessors (not part of the reference's object model; the shim uses them when present)
    def point3D_track_lengths(self, ids):
        return self._store.tlen[self._store.rows(ids)]

    def set_point3D_coordinates(self, ids, xyz):
        self._store.xyz[self._store.rows(ids)] = xyz

    def project_image_3d_points(self, imid, pts3dids=None):
        """(pts2dids, pts3dids, kps, depth, success) — reference points3D_utils.py:9-25 + geometry.py:13-19."""
        image = self.images[imid]
        pts2dids = None
        if pts3dids is None:
            pts2dids = image.get_observation_point2D_idxs()
            pts3dids = image.point3D_ids(pts2dids)
            if len(pts3dids) == 0:
                return None, None, None, None, False
        kps, depth = project3D_colmap(image, self.rec.cameras[image.camera_id], self.point3D_coordinates(pts3dids))
        return pts2dids, pts3dids, kps, depth, True

    def lifted_pointcovs_cam(self, dd, camera, keypoints, var_d, sigma_q=1):
        """Camera-frame covariance of points lifted from depth dd at `keypoints`: depth variance along the
        viewing ray plus pixel noise sigma_q in the image plane (reference points3D_utils.py:27-48)."""
        ff_inv = 1.0 / np.array([camera.focal_length_x, camera.focal_length_y])
        cc = np.array([camera.principal_point_x, camera.principal_point_y])
        ray = np.concatenate([(keypoints - cc) * ff_inv, np.ones((keypoints.shape[0], 1))], axis=1)
        cov = var_d[:, None, None] * ray[:, :, None] * ray[:, None, :]
        px = np.clip(dd[:, None] * ff_inv[None, :], -1e6, 1e6) ** 2 * sigma_q**2
        cov[:, 0, 0] += px[:, 0]
        cov[:, 1, 1] += px[:, 1]
        return cov

    def rotate_covs(self, Covs, R):
        return R[None] @ Covs @ R.T[None]

    def rotate_covs_to_world(self, Covs, imid):
        return self.rotate_covs(Covs, self.images[imid].cam_from_world.rotation.matrix())

    def rotate_covs_to_cam(self, Covs_world, imid):
        return self.rotate_covs(Covs_world, self.images[imid].cam_from_world.rotation.matrix().T)

    # -- depth bookkeeping of the reference's DepthUtils mixin (reconstruction/mixins/depth_utils.py:52-92) ------
    def activate_depths(self, imids):
        for imid in imids:
            d = self.images[imid].depth
            if not d.activated:
                d.activated = True
                d.data = d.data_prior.copy()

    def rescale_all(self, shift_scales):
        for imid, (shift, scale) in shift_scales.items():
            d = self.images[imid].depth
            d.data_prior = d.data_prior * scale + shift
            d.scale *= scale
            d.shift = d.shift * scale + shift
            d.uncertainty = d.uncertainty * scale**2
        for imid, (shift, scale) in shift_scales.items():
            d = self.images[imid].depth
            d.uncertainty_update = d.uncertainty_update * scale**2

    def reg_image_ids(self):
        return [i for i, im in self.images.items() if im.has_pose]

    def find_points3D_with_small_triangulation_angle(self, min_angle, point3D_ids):
        return np.array(self.obs.find_small_angle_points_mask(float(min_angle), point3D_ids))

    def find_local_bundle_ids(self, imid, num_images):
        """Images sharing the most 3-D points with imid (stand-in for COLMAP's FindLocalBundle)."""
        mine = set(self.images[imid].point3D_ids(self.images[imid].get_observation_point2D_idxs()))
        scores = []
        for other, im in self.images.items():
            if other == imid or not im.has_pose:
                continue
            shared = len(mine & set(im.point3D_ids(im.get_observation_point2D_idxs())))
            if shared:
                scores.append((-shared, other))
        return [o for _, o in sorted(scores)[:num_images]]


def scene_from_problem(prob, truth=None, map_size=(129, 97), image_size=(1600.0, 1200.0), seed=0,
                       prior_noise=0.0263, first_image_id=1, with_points=True):
    """Builds a NumpyReconstruction from a synthetic BAProblem (mpsfm_amd.synthetic.make_scene):
    one keypoint per observation, per-image prior depth maps splatted from the true camera-frame
    depths (so the sampled priors are roughly right), variance map from the reference's model."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    scene = NumpyReconstruction()
    W, H = map_size
    n_cams = prob.n_cams
    tq = truth["cam_quat"] if truth is not None else prob.cam_quat
    tt = truth["cam_t"] if truth is not None else prob.cam_t
    tX = truth["pts"] if truth is not None else prob.pts
    for k in range(prob.cam_intr.shape[0]):
        scene.rec.cameras[k] = NumpyCamera(k, prob.cam_intr[k], image_size[0], image_size[1], W, H)
    order = np.argsort(prob.obs_cam, kind="stable")
    starts = np.searchsorted(prob.obs_cam[order], np.arange(n_cams + 1))
    Rt = R_from_quat(tq)
    imids = [first_image_id + c for c in range(n_cams)]
    for c in range(n_cams):
        idx = order[starts[c]:starts[c + 1]]
        kps = prob.obs_xy[idx]
        cam = scene.rec.cameras[int(prob.cam_intr_idx[c])]
        ztrue = (tX[prob.obs_pt[idx]] @ Rt[c].T + tt[c])[:, 2]
        depth = None
        if len(idx) >= 3:
            gy, gx = np.mgrid[0:H, 0:W]
            tree = cKDTree(np.stack([kps[:, 0] * cam.sx, kps[:, 1] * cam.sy], 1))
            _, nn = tree.query(np.stack([gx.ravel(), gy.ravel()], 1), k=min(3, len(idx)))
            nn = nn.reshape(H * W, -1)
            dmap = ztrue[nn].mean(axis=1).reshape(H, W) * np.exp(rng.normal(0, prior_noise, (H, W)))
            var = np.maximum((prior_noise * dmap) ** 2, 0.02**2)
            valid = (rng.uniform(size=(H, W)) > 0.03).astype(np.float64)
            depth = NumpyDepth(dmap, var, valid, cam, kps)
        img = NumpyImage(imids[c], int(prob.cam_intr_idx[c]), Rigid3d(prob.cam_quat[c], prob.cam_t[c]), kps, 1.0, depth)
        if depth is None:
            img.depth = NumpyDepth(np.ones((H, W)), np.ones((H, W)), np.zeros((H, W)), cam, kps)
            img.depth.activated = False
        img._obs_index = idx
        scene.images[imids[c]] = img
    # points and tracks
    pt_order = np.argsort(prob.obs_pt, kind="stable")
    pstarts = np.searchsorted(prob.obs_pt[pt_order], np.arange(prob.n_pts + 1))
    local_idx = np.empty(prob.n_obs, dtype=np.int64)
    for c in range(n_cams):
        idx = order[starts[c]:starts[c + 1]]
        local_idx[idx] = np.arange(len(idx))
    scene._obs_image = np.array([imids[int(c)] for c in prob.obs_cam], np.int64)  # observation -> (image id, point2D idx)
    scene._obs_point2D = local_idx
    pid_of = {}
    for p in range(prob.n_pts if with_points else 0):
        obs = pt_order[pstarts[p]:pstarts[p + 1]]
        if len(obs) == 0:
            continue
        tr = Track()
        for o in obs:
            tr.add_element(imids[int(prob.obs_cam[o])], int(local_idx[o]))
        pid_of[p] = scene.obs.add_point3D(prob.pts[p], tr)
    scene._pid_of_problem_point = pid_of
    return scene


def correspondences_from_problem(scene, prob, false_matches=0, seed=0):
    """A correspondence graph for a scene built by scene_from_problem: every pair of observations of one landmark is a
    match (what exhaustive matching + geometric verification would find), plus `false_matches` random wrong ones."""
    cg = NumpyCorrespondenceGraph()
    for imid, im in scene.images.items():
        cg.add_image(imid, len(im.kps))
    order = np.argsort(prob.obs_pt, kind="stable")
    start = np.searchsorted(prob.obs_pt[order], np.arange(prob.n_pts + 1))
    im_of, kp_of = scene._obs_image[order], scene._obs_point2D[order]
    length = np.diff(start)
    pairs = {}
    for L in np.unique(length):
        if L < 2:
            continue
        pts = np.flatnonzero(length == L)
        base = start[pts][:, None] + np.arange(L)[None, :]          # [n, L] observation slots of these landmarks
        for a in range(L):
            for b in range(a + 1, L):
                i1, i2 = im_of[base[:, a]], im_of[base[:, b]]
                k1, k2 = kp_of[base[:, a]], kp_of[base[:, b]]
                swap = i1 > i2
                i1, i2, k1, k2 = np.where(swap, i2, i1), np.where(swap, i1, i2), np.where(swap, k2, k1), np.where(swap, k1, k2)
                key = i1 * (1 << 32) + i2
                for kk in np.unique(key):
                    sel = key == kk
                    pairs.setdefault((int(kk >> 32), int(kk & 0xffffffff)), []).append(np.stack([k1[sel], k2[sel]], 1))
    rng = np.random.default_rng(seed)
    ids = sorted(scene.images)
    for _ in range(false_matches):
        a, b = sorted(rng.choice(len(ids), 2, replace=False))
        ia, ib = ids[a], ids[b]
        pairs.setdefault((ia, ib), []).append(np.array([[rng.integers(len(scene.images[ia].kps)), rng.integers(len(scene.images[ib].kps))]]))
    for (ia, ib), ms in pairs.items():
        if ia == ib:
            continue
        cg.add_correspondences(ia, ib, np.unique(np.concatenate(ms), axis=0))
    cg.finalize()
    return cg
