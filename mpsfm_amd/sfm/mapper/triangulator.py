"""``MpsfmTriangulator`` with the reference's interface (mpsfm/sfm/mapper/triangulator.py:16-175).

The reference delegates track finding / merging / completion to ``pycolmap.IncrementalTriangulator``
(C++ graph logic — SURVEY.md §8f row f2, not part of this round) and adds, in Python, the
replacement of low-parallax points by depth-lifted points (:49-83, :125-161).  Here the per-track
numerics (linear multi-view triangulation, max pairwise triangulation angle, reprojection error,
cheirality) run as batch HIP kernels through the C ABI (``mpsfm_triangulate_tracks`` /
``mpsfm_filter_tracks``); the graph engine is pluggable (anything with the IncrementalTriangulator
method names, e.g. the real pycolmap object).
"""

from __future__ import annotations

import numpy as np

from ...baseclass import BaseClass
from ...problem import Tracks
from ...utils.geometry import has_point_positive_depth
from .bundle_adjustment import pinhole_params


def tracks_from_scene(scene, point3D_ids, keep_elements=False) -> tuple[Tracks, list]:
    """CSR tracks of the given points over the images of a scene (cameras indexed by image order).  With
    `keep_elements` the returned Tracks carries `.elements`: (image_id, point2D_idx) per track element."""
    imids = sorted(scene.images.keys())
    cam_of = {imid: i for i, imid in enumerate(imids)}
    cam_ids = [scene.images[i].camera_id for i in imids]
    uniq = sorted(set(cam_ids))
    start, el_cam, el_xy, elements = [0], [], [], []
    for pid in point3D_ids:
        for el in scene.points3D[int(pid)].track.elements:
            el_cam.append(cam_of[el.image_id])
            el_xy.append(np.asarray(scene.images[el.image_id].points2D[el.point2D_idx].xy, np.float64))
            if keep_elements:
                elements.append((int(el.image_id), int(el.point2D_idx)))
        start.append(len(el_cam))
    tr = Tracks(
        cam_quat=np.array([scene.images[i].cam_from_world.rotation.quat for i in imids]).reshape(-1, 4),
        cam_t=np.array([scene.images[i].cam_from_world.translation for i in imids]).reshape(-1, 3),
        cam_intr=np.array([pinhole_params(scene.rec.cameras[c]) for c in uniq]).reshape(-1, 4),
        cam_intr_idx=np.array([uniq.index(c) for c in cam_ids], np.int32),
        track_start=np.array(start, np.int64), el_cam=np.array(el_cam, np.int32),
        el_xy=np.array(el_xy, np.float64).reshape(-1, 2),
    )
    tr.elements = elements
    return tr, imids


def track_quality(scene, point3D_ids, device: int = 0):
    """(max triangulation angle [rad] per point, squared reprojection error and in-front flag per
    track element) at the points' current coordinates — HIP batch kernel."""
    from ... import capi

    if len(point3D_ids) == 0:
        return np.zeros(0), np.zeros(0), np.zeros(0, bool)
    tr, _ = tracks_from_scene(scene, point3D_ids)
    xyz = scene.point3D_coordinates(point3D_ids)
    return capi.filter_tracks(tr, xyz, device)


def triangulate_points(scene, point3D_ids, device: int = 0) -> np.ndarray:
    """Linear multi-view triangulation of the tracks of the given points (HIP batch kernel)."""
    from ... import capi

    tr, _ = tracks_from_scene(scene, point3D_ids)
    return capi.triangulate_tracks(tr, device)


_TRI_OPTION_KEYS = ("max_transitivity", "create_max_angle_error", "continue_max_angle_error", "merge_max_reproj_error",
                    "complete_max_reproj_error", "complete_max_transitivity", "re_max_angle_error", "re_min_ratio", "re_max_trials",
                    "min_angle", "ignore_two_view_tracks")


class ColmapTriangulatorWrapper:
    def __getattr__(self, name):
        if name in self.__dict__:
            return self.__dict__[name]
        return getattr(self._triangulator, name)


class MpsfmTriangulator(BaseClass, ColmapTriangulatorWrapper):
    """MP-SfM triangulator wrapper (API of the reference's class of the same name)."""

    default_conf = {
        "hard_angle": 1.5,
        "colmap_options": "<--->",
        "new_retry_nbatch": 5,
        "re_ignore_two_view_tracks": False,
        "retri_min_angle": 1.5,
        "lift_low_parallax": True,
        "nsafe_threshold": 60,
        "verbose": 0,
    }

    def _init(self, mpsfm_rec, correspondences_graph=None, engine=None, **kwargs):
        self.mpsfm_rec = mpsfm_rec
        self.correspondences_graph = correspondences_graph
        # the track-graph engine: anything with pycolmap.IncrementalTriangulator's method names; by default the native
        # one of this repository (built on first use: the scene's keypoints must all be there by then)
        self._triangulator = engine
        self.device = int(kwargs.get("device", 0))
        opts = self.conf.colmap_options
        if isinstance(opts, dict):
            self.options = dict(opts)
        elif opts is None or isinstance(opts, str):
            self.options = {}
        else:  # an OmegaConf node or a pycolmap options object: its triangulator fields (reference :34-40)
            self.options = {k: (opts[k] if hasattr(opts, "__getitem__") else getattr(opts, k)) for k in _TRI_OPTION_KEYS
                            if (k in opts if hasattr(opts, "__contains__") else hasattr(opts, k))}

    def _require_engine(self):
        if self._triangulator is None:
            if self.correspondences_graph is None:
                raise ValueError("MpsfmTriangulator needs the correspondence graph (reference mapper/base.py:183) or an `engine`")
            from .track_engine import HipIncrementalTriangulator

            self._triangulator = HipIncrementalTriangulator(self.correspondences_graph, self.mpsfm_rec, self.device)

    # -- depth lifting of low-parallax points (reference :49-83 and :125-161) ----------------------
    def _lift_points(self, point3D_ids):
        rec = self.mpsfm_rec
        new_ids = []
        for point3D_id in point3D_ids:
            point3D = rec.points3D[int(point3D_id)]
            imids = [el.image_id for el in point3D.track.elements]
            ptids = [el.point2D_idx for el in point3D.track.elements]
            cams_from_world = [rec.images[i].cam_from_world for i in imids]
            rec.obs.delete_point3D(int(point3D_id))
            for liftid, limid in enumerate(imids):
                lift_image = rec.images[limid]
                if not lift_image.depth.activated:
                    continue
                xy = np.array([lift_image.points2D[ptids[liftid]].xy])
                if not lift_image.depth.valid_at_kps(xy)[0]:
                    continue
                d = lift_image.depth.data_at_kps(xy)[:, None]
                cam = rec.rec.cameras[lift_image.camera_id]
                xyz = lift_image.cam_from_world.inverse() * (np.concatenate([cam.cam_from_img(xy), np.ones((1, 1))], -1) * d)
                track = type(point3D.track)()
                for imid_, ptid, cfw in zip(imids, ptids, cams_from_world):
                    if has_point_positive_depth(cfw.matrix(), xyz[0]):
                        track.add_element(imid_, ptid)
                new_ids.append(rec.obs.add_point3D(xyz[0], track))
                break
        return new_ids

    def lift_low_parallax(self, point3D_ids, min_angle):
        """Replace the points among `point3D_ids` whose largest triangulation angle is below
        `min_angle` degrees by points lifted from the first activated depth map of their track
        (reference :49-83: the mask of mpsfm_rec.find_points3D_with_small_triangulation_angle, here
        taken from the HIP batch kernel directly)."""
        ids = np.array([int(p) for p in point3D_ids if int(p) in self.mpsfm_rec.points3D], dtype=np.int64)
        if len(ids) == 0:
            return []
        ang, _, _ = track_quality(self.mpsfm_rec, ids, self.device)
        return self._lift_points(ids[ang < np.deg2rad(float(min_angle))])

    def triangulate_image(self, imid, **kwargs) -> bool:
        self._require_engine()
        in3D = set(self.mpsfm_rec.points3D.keys())
        self._triangulator.triangulate_image(self.options, imid)
        if self.conf.lift_low_parallax:
            self.lift_low_parallax(set(self.mpsfm_rec.points3D.keys()) - in3D, self.conf.hard_angle)
        return True

    def retriangulate(self):
        self._require_engine()
        rec = self.mpsfm_rec
        risky_imids = []
        if self.conf.new_retry_nbatch is not None:
            for imid in list(rec.registered_images):
                image = rec.images[imid]
                p3d = set(image.point3D_ids(image.get_observation_point2D_idxs()))
                nsafe = sum(1 for p in p3d if len(rec.points3D[p].track.elements) > 2)
                if nsafe < self.conf.nsafe_threshold:
                    risky_imids.append(imid)
            expanded = sum((rec.find_local_bundle_ids(i, self.conf.new_retry_nbatch) for i in risky_imids), [])
            risky_imids = risky_imids + expanded
        out = self._triangulator.retriangulate(self.options, set(risky_imids))
        self.lift_low_parallax(list(rec.points3D.keys()), self.conf.retri_min_angle)
        return out

    def complete_image(self, imid):
        self._require_engine()
        return self._triangulator.complete_image(self.options, imid)

    def complete_all_tracks(self):
        self._require_engine()
        return self._triangulator.complete_all_tracks(self.options)

    def complete_tracks(self, points3D):
        self._require_engine()
        return self._triangulator.complete_tracks(self.options, points3D)

    def merge_tracks(self, points3D):
        self._require_engine()
        return self._triangulator.merge_tracks(self.options, points3D)

    def merge_all_tracks(self):
        self._require_engine()
        return self._triangulator.merge_all_tracks(self.options)

    def complete_and_merge_all_tracks(self) -> int:
        return self.complete_all_tracks() + self.merge_all_tracks()

    def complete_and_merge_tracks(self, points3D) -> int:
        return self.complete_tracks(points3D) + self.merge_tracks(points3D)
