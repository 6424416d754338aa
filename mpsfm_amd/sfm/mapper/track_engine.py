"""``IncrementalTriangulator``-shaped engine over libmpsfm_hip.

The reference builds ``pycolmap.IncrementalTriangulator(correspondence_graph, reconstruction, obs_manager)`` and calls
``triangulate_image(options, image_id)``, ``complete_image``, ``complete_tracks / complete_all_tracks``,
``merge_tracks / merge_all_tracks`` and the fork's ``retriangulate(options, ignore_image_ids)`` on it
(mpsfm/sfm/mapper/triangulator.py:32-48, 88-100, 123).  ``HipIncrementalTriangulator`` has the same method names and
return values (counts as COLMAP defines them).  The track-graph walk runs in the native library
(csrc/triangulator.hip: COLMAP 3.11 semantics, candidate tracks estimated in one GPU batch per call); the scene stays
where it is: before a call the poses / registration flags / keypoint -> point assignment are handed over, after it the
operation log (add point, add observation, delete point) is replayed on ``mpsfm_rec.obs`` — the reference's
ObservationManager keeps doing the bookkeeping.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from ... import capi
from .bundle_adjustment import pinhole_params

_OPT_FIELDS = ("max_transitivity", "create_max_angle_error", "continue_max_angle_error", "merge_max_reproj_error",
               "complete_max_reproj_error", "complete_max_transitivity", "re_max_angle_error", "re_min_ratio", "re_max_trials",
               "min_angle", "ignore_two_view_tracks")


class CTriOptions(C.Structure):
    _fields_ = [("max_transitivity", C.c_int32), ("create_max_angle_error", C.c_double), ("continue_max_angle_error", C.c_double),
                ("merge_max_reproj_error", C.c_double), ("complete_max_reproj_error", C.c_double), ("complete_max_transitivity", C.c_int32),
                ("re_max_angle_error", C.c_double), ("re_min_ratio", C.c_double), ("re_max_trials", C.c_int32), ("min_angle", C.c_double),
                ("ignore_two_view_tracks", C.c_int32)]


class CTriGraph(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("kp_start", C.c_void_p), ("kp_xy", C.c_void_p), ("cam_intr", C.c_void_p),
                ("corr_start", C.c_void_p), ("corr_kp", C.c_void_p)]


class CTriState(C.Structure):
    _fields_ = [("registered", C.c_void_p), ("cam_quat_xyzw", C.c_void_p), ("cam_t", C.c_void_p), ("kp_point", C.c_void_p),
                ("n_points", C.c_int64), ("xyz", C.c_void_p)]


def tri_options(options) -> CTriOptions:
    """pycolmap.IncrementalTriangulatorOptions (or a dict of its fields) -> the C struct; missing fields = COLMAP defaults."""
    o = CTriOptions()
    capi.lib().mpsfm_tri_default_options(C.byref(o))
    for k in _OPT_FIELDS:
        v = options.get(k) if isinstance(options, dict) else getattr(options, k, None)
        if v is not None:
            setattr(o, k, int(v) if k in ("max_transitivity", "complete_max_transitivity", "re_max_trials", "ignore_two_view_tracks") else float(v))
    if o.max_transitivity != 1:
        raise NotImplementedError("only max_transitivity = 1 (COLMAP's default) is implemented")
    return o


def _image_pairs(cg):
    if hasattr(cg, "image_pairs"):
        return list(cg.image_pairs())
    import pycolmap  # the real correspondence graph

    return [tuple(pycolmap.pair_id_to_image_pair(pid)) for pid in cg.num_correspondences_between_all_images()]


def graph_arrays(correspondence_graph, mpsfm_rec):
    """Keypoints and correspondence graph of a scene as the flat arrays of `mpsfm_tri_graph` (include/mpsfm_hip.h)."""
    image_ids = sorted(mpsfm_rec.images.keys())
    im_index = {imid: i for i, imid in enumerate(image_ids)}
    nkp = [len(mpsfm_rec.images[i].points2D) for i in image_ids]
    kp_start = np.concatenate([[0], np.cumsum(nkp)]).astype(np.int64)
    kp_xy = np.concatenate([np.asarray(mpsfm_rec.keypoints(i), np.float64).reshape(-1, 2) for i in image_ids]) if sum(nkp) else np.zeros((0, 2))
    intr = np.array([pinhole_params(mpsfm_rec.rec.cameras[mpsfm_rec.images[i].camera_id]) for i in image_ids], np.float64).reshape(-1, 4)
    src, dst = [], []
    for id1, id2 in _image_pairs(correspondence_graph):
        if id1 not in im_index or id2 not in im_index:
            continue
        m = np.asarray(correspondence_graph.find_correspondences_between_images(id1, id2), np.int64).reshape(-1, 2)
        a, b = kp_start[im_index[id1]] + m[:, 0], kp_start[im_index[id2]] + m[:, 1]
        src += [a, b]
        dst += [b, a]
    n_kp = int(kp_start[-1])
    if src:
        src, dst = np.concatenate(src), np.concatenate(dst)
        order = np.lexsort((dst, src))
        src, dst = src[order], dst[order]
    else:
        src = dst = np.zeros(0, np.int64)
    return dict(image_ids=image_ids, im_index=im_index, kp_start=kp_start, kp_xy=np.ascontiguousarray(kp_xy), intr=np.ascontiguousarray(intr),
                corr_start=np.searchsorted(src, np.arange(n_kp + 1)).astype(np.int64), corr_kp=np.ascontiguousarray(dst, np.int64),
                kp_image=np.repeat(np.arange(len(image_ids)), nkp))


def state_arrays(mpsfm_rec, image_ids, kp_start):
    """The mutable part of a scene as the arrays of `mpsfm_tri_state`; also the scene ids of the listed points."""
    n_im = len(image_ids)
    reg = np.zeros(n_im, np.uint8)
    quat, trans = np.zeros((n_im, 4)), np.zeros((n_im, 3))
    quat[:, 3] = 1.0
    kp_pid = np.full(int(kp_start[-1]), -1, np.int64)
    for i, imid in enumerate(image_ids):
        im = mpsfm_rec.images[imid]
        reg[i] = 1 if im.has_pose else 0
        if im.has_pose:
            quat[i], trans[i] = im.cam_from_world.rotation.quat, im.cam_from_world.translation
        idx = np.asarray(im.get_observation_point2D_idxs(), np.int64)
        if len(idx):
            kp_pid[kp_start[i] + idx] = np.asarray(im.point3D_ids(idx), dtype=np.uint64).astype(np.int64)
    point_ids = np.unique(kp_pid[kp_pid >= 0])
    kp_point = np.where(kp_pid >= 0, np.searchsorted(point_ids, kp_pid), -1).astype(np.int64)
    xyz = np.asarray(mpsfm_rec.point3D_coordinates(point_ids), np.float64).reshape(-1, 3) if len(point_ids) else np.zeros((0, 3))
    return dict(registered=reg, cam_quat_xyzw=quat, cam_t=trans, kp_point=kp_point, xyz=np.ascontiguousarray(xyz)), point_ids


class HipIncrementalTriangulator:
    def __init__(self, correspondence_graph, mpsfm_rec, device: int = 0):
        L = capi.lib()
        self.rec, self.cg, self.device = mpsfm_rec, correspondence_graph, device
        ga = graph_arrays(correspondence_graph, mpsfm_rec)
        for k, v in ga.items():
            setattr(self, k, v)
        g = CTriGraph(len(self.image_ids), self.kp_start.ctypes.data, self.kp_xy.ctypes.data, self.intr.ctypes.data,
                      self.corr_start.ctypes.data, self.corr_kp.ctypes.data)
        self._image_ids_arr = np.asarray(self.image_ids)
        self._h = C.c_void_p(None)
        L.mpsfm_triangulator_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        capi._check(L.mpsfm_triangulator_create(C.byref(g), device, C.byref(self._h)))
        capi._track(self)  # closed before the interpreter's exit handlers give way to the C runtime's
        for name in ("triangulate_image", "complete_image"):
            getattr(L, "mpsfm_triangulator_" + name).argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        for name in ("complete_tracks", "merge_tracks"):
            getattr(L, "mpsfm_triangulator_" + name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.mpsfm_triangulator_retriangulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        L.mpsfm_triangulator_set_state.argtypes = [C.c_void_p, C.c_void_p]
        L.mpsfm_triangulator_get_ops.argtypes = [C.c_void_p] * 5
        L.mpsfm_triangulator_get_op_elements.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("num_ops", "num_points", "num_op_elements"):
            getattr(L, "mpsfm_triangulator_" + name).argtypes = [C.c_void_p]
            getattr(L, "mpsfm_triangulator_" + name).restype = C.c_int64
        L.mpsfm_triangulator_destroy.argtypes = [C.c_void_p]
        L.mpsfm_triangulator_destroy.restype = None
        L.mpsfm_triangulator_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
        self.last_ops = None

    def close(self):
        if self._h:
            capi.lib().mpsfm_triangulator_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- state hand-over and replay ------------------------------------------------------------------------------------
    def _sync(self):
        st, self._point_ids = state_arrays(self.rec, self.image_ids, self.kp_start)
        self.last_state = st  # what the engine was handed
        cst = CTriState(st["registered"].ctypes.data, st["cam_quat_xyzw"].ctypes.data, st["cam_t"].ctypes.data, st["kp_point"].ctypes.data,
                        len(self._point_ids), st["xyz"].ctypes.data)
        capi._check(capi.lib().mpsfm_triangulator_set_state(self._h, C.byref(cst)))
        self._scene_id = dict(enumerate(self._point_ids.tolist()))  # engine point -> scene point id

    def _track_types(self):
        rec = self.rec
        if hasattr(rec, "Track"):
            return rec.Track, rec.TrackElement
        import pycolmap

        return pycolmap.Track, pycolmap.TrackElement

    def _element(self, kp):
        i = int(self.kp_image[kp])
        return self.image_ids[i], int(kp - self.kp_start[i])

    def _replay(self):
        L = capi.lib()
        n = int(L.mpsfm_triangulator_num_ops(self._h))
        typ, a, b, xyz = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros((n, 3))
        els = np.zeros(int(L.mpsfm_triangulator_num_op_elements(self._h)), np.int64)
        if n:
            capi._check(L.mpsfm_triangulator_get_ops(self._h, typ.ctypes.data, a.ctypes.data, b.ctypes.data, xyz.ctypes.data))
        if len(els):
            capi._check(L.mpsfm_triangulator_get_op_elements(self._h, els.ctypes.data))
        Track, TrackElement = self._track_types()
        obs, sid = self.rec.obs, self._scene_id
        # plain Python ints for the loop below (NumPy scalars cost ~10 x as much per use): the (image id, keypoint index) of every
        # element of the added points and of every added observation, the op fields
        ids_arr = self._image_ids_arr
        el_im = self.kp_image[els] if len(els) else np.zeros(0, np.int64)
        el_image = ids_arr[el_im].tolist()
        el_idx = (els - self.kp_start[el_im]).tolist()
        is_obs = typ == 1
        ob_im = self.kp_image[np.where(is_obs, b, 0)]
        ob_image = ids_arr[ob_im].tolist()
        ob_idx = np.where(is_obs, b - self.kp_start[ob_im], 0).tolist()
        typ_l, a_l, b_l = typ.tolist(), a.tolist(), b.tolist()
        e0 = 0
        modified = set()
        for k in range(n):
            tk, ak = typ_l[k], a_l[k]
            if tk == 0:  # add point
                tr = Track()
                e1 = e0 + b_l[k]
                for j in range(e0, e1):
                    tr.add_element(el_image[j], el_idx[j])
                e0 = e1
                pid = int(obs.add_point3D(xyz[k].copy(), tr))
                sid[ak] = pid
                modified.add(pid)
            elif tk == 1:  # add observation
                pid = sid[ak]
                obs.add_observation(pid, TrackElement(ob_image[k], ob_idx[k]))
                modified.add(pid)
            else:  # delete point
                pid = sid[ak]
                obs.delete_point3D(pid)
                modified.discard(pid)
        self.last_ops = dict(type=typ, a=a, b=b, xyz=xyz, elements=els)
        self.modified_point3D_ids = modified
        return n

    def _engine_points(self, point3D_ids):
        pos = {int(p): i for i, p in enumerate(self._point_ids)}
        return np.array([pos[int(p)] for p in point3D_ids if int(p) in pos], np.int64)

    def stats(self):
        v = [C.c_int64(0) for _ in range(3)]
        capi._check(capi.lib().mpsfm_triangulator_stats(self._h, *[C.byref(x) for x in v]))
        return dict(batch_candidates=v[0].value, batch_hits=v[1].value, host_estimates=v[2].value)

    # -- pycolmap.IncrementalTriangulator's methods ----------------------------------------------------------------------
    def _image_call(self, fn, options, image_id):
        o = tri_options(options)
        self._sync()
        cnt = C.c_int64(0)
        capi._check(fn(self._h, C.byref(o), self.im_index[image_id], C.byref(cnt)))
        self._replay()
        return int(cnt.value)

    def triangulate_image(self, options, image_id):
        return self._image_call(capi.lib().mpsfm_triangulator_triangulate_image, options, image_id)

    def complete_image(self, options, image_id):
        return self._image_call(capi.lib().mpsfm_triangulator_complete_image, options, image_id)

    def _tracks_call(self, fn, options, point3D_ids):
        o = tri_options(options)
        self._sync()
        cnt = C.c_int64(0)
        if point3D_ids is None:
            capi._check(fn(self._h, C.byref(o), None, -1, C.byref(cnt)))
        else:
            ids = self._engine_points(point3D_ids)
            capi._check(fn(self._h, C.byref(o), ids.ctypes.data if len(ids) else None, len(ids), C.byref(cnt)))
        self._replay()
        return int(cnt.value)

    def complete_tracks(self, options, point3D_ids):
        return self._tracks_call(capi.lib().mpsfm_triangulator_complete_tracks, options, point3D_ids)

    def complete_all_tracks(self, options):
        return self._tracks_call(capi.lib().mpsfm_triangulator_complete_tracks, options, None)

    def merge_tracks(self, options, point3D_ids):
        return self._tracks_call(capi.lib().mpsfm_triangulator_merge_tracks, options, point3D_ids)

    def merge_all_tracks(self, options):
        return self._tracks_call(capi.lib().mpsfm_triangulator_merge_tracks, options, None)

    def retriangulate(self, options, ignore_image_ids=()):
        o = tri_options(options)
        self._sync()
        ig = np.array([self.im_index[i] for i in ignore_image_ids if i in self.im_index], np.int32)
        cnt = C.c_int64(0)
        capi._check(capi.lib().mpsfm_triangulator_retriangulate(self._h, C.byref(o), ig.ctypes.data if len(ig) else None, len(ig), C.byref(cnt)))
        self._replay()
        return int(cnt.value)
