"""``ObservationManager``-shaped front whose per-track arithmetic runs in the HIP kernels.

The reference keeps ``mpsfm_rec.obs = pycolmap.ObservationManager(rec, cg)`` (mapper/base.py:179) and calls, around
every bundle adjustment (mapper/base.py:686-797) and from the triangulator (points3D_utils.py:64-71):

    obs.filter_observations_with_negative_depth()
    obs.filter_points3D(max_reproj_error, min_tri_angle, point3D_ids)        # fork: ids as third argument
    obs.filter_all_points3D(max_reproj_error, min_tri_angle)
    obs.find_small_angle_points_mask(min_angle_deg, point3D_ids)             # fork only

All four are loops of per-track numerics (squared reprojection error with COLMAP's cheirality rule, largest pairwise
triangulation angle) followed by bookkeeping (delete an observation / a point).  ``HipObservationManager`` gathers the
tracks through the accessor names SURVEY.md §8b lists, runs ONE ``mpsfm_filter_tracks`` launch per filter and applies
the deletions through the wrapped manager's own ``delete_point3D`` / ``delete_observation`` — so the real pycolmap
object keeps its internal state consistent.  Every other attribute (``add_point3D``, ``filter_images``,
``deregister_image``, ``num_visible_points3D`` …) is passed through.

Decision rules follow COLMAP 3.11 ``ObservationManager`` (the fork's C++ is not in the reference tree — parity
unpinned; the rules are frozen by tests/test_gpu_seam.py against oracle/tri_oracle.c numerics):
  * an element is bad when the point is not in front of its camera (z < eps) or its squared reprojection error exceeds
    max_reproj_error^2; a point of track length < 2, or with at most one good element left, is deleted whole, otherwise
    its bad observations are deleted (deleting from a 2-element track deletes the point);
  * then a point none of whose camera pairs reaches min_tri_angle is deleted;
  * the return value counts deleted observations in the first step (+ whole track lengths) and points in the second.
"""

from __future__ import annotations

import numpy as np


def _default_numerics(tracks, xyz, device):
    from ... import capi

    return capi.filter_tracks(tracks, xyz, device)


def reprojection_decisions(track_start, sq_err, front, max_reproj_error):
    """Per track: (delete_whole [T] bool, bad element mask [E] bool) from the kernel outputs."""
    start = np.asarray(track_start, np.int64)
    bad = (~np.asarray(front, bool)) | (np.asarray(sq_err) > max_reproj_error * max_reproj_error)
    length = np.diff(start)
    nbad = np.add.reduceat(np.concatenate([bad.astype(np.int64), [0]]), start[:-1])[: len(length)] if len(length) else np.zeros(0, np.int64)
    nbad = np.where(length > 0, nbad, 0)
    whole = (length < 2) | (nbad >= length - 1)
    return whole, bad


class HipObservationManager:
    def __init__(self, mpsfm_rec, obs=None, device: int = 0, numerics=None):
        self.__dict__["mpsfm_rec"] = mpsfm_rec
        self.__dict__["_obs"] = obs if obs is not None else mpsfm_rec.obs
        self.__dict__["device"] = device
        self.__dict__["_numerics"] = numerics or _default_numerics  # tests inject the oracle's numerics here

    def __getattr__(self, name):  # bookkeeping stays with the wrapped manager
        if name.startswith("__") or "_obs" not in self.__dict__:  # copy / pickle probes on a half-built object
            raise AttributeError(name)
        return getattr(self.__dict__["_obs"], name)

    # -- gathering -------------------------------------------------------------------------------
    def _tracks(self, point3D_ids):
        from ..mapper.triangulator import tracks_from_scene

        rec = self.mpsfm_rec
        ids = [int(p) for p in point3D_ids if int(p) in rec.points3D]
        tr, _ = tracks_from_scene(rec, ids, keep_elements=True)
        return ids, tr

    # -- the fork's mask -------------------------------------------------------------------------
    def find_small_angle_points_mask(self, min_angle_deg, point3D_ids):
        """True where the largest pairwise triangulation angle of the point's track is below min_angle_deg
        (reference points3D_utils.py:64-71; one mask entry per given id, in order)."""
        ids = [int(p) for p in point3D_ids]
        if not ids:
            return np.zeros(0, bool)
        _, tr = self._tracks(ids)
        ang, _, _ = self._numerics(tr, self.mpsfm_rec.point3D_coordinates(ids), self.device)
        return ang < np.deg2rad(float(min_angle_deg))

    # -- filters ---------------------------------------------------------------------------------
    def _filter_reprojection(self, max_reproj_error, ids):
        rec, obs = self.mpsfm_rec, self._obs
        ids, tr = self._tracks(ids)
        if not ids:
            return 0
        _, err, front = self._numerics(tr, rec.point3D_coordinates(ids), self.device)
        whole, bad = reprojection_decisions(tr.track_start, err, front, float(max_reproj_error))
        start = tr.track_start
        num = 0
        for k, pid in enumerate(ids):
            e0, e1 = int(start[k]), int(start[k + 1])
            if whole[k]:
                num += e1 - e0
                obs.delete_point3D(pid)
                continue
            els = [tr.elements[e] for e in range(e0, e1) if bad[e]]
            num += len(els)
            for image_id, point2D_idx in els:
                obs.delete_observation(image_id, point2D_idx)
            if hasattr(rec.points3D[pid], "error"):
                good = ~bad[e0:e1]
                rec.points3D[pid].error = float(np.sqrt(err[e0:e1][good]).sum() / (e1 - e0 - len(els)))
        return num

    def _filter_small_angles(self, min_tri_angle, ids):
        rec, obs = self.mpsfm_rec, self._obs
        ids, tr = self._tracks(ids)
        if not ids:
            return 0
        ang, _, _ = self._numerics(tr, rec.point3D_coordinates(ids), self.device)
        drop = ~(ang >= np.deg2rad(float(min_tri_angle)))
        for pid in np.asarray(ids, dtype=object)[drop]:
            obs.delete_point3D(int(pid))
        return int(drop.sum())

    def filter_points3D(self, max_reproj_error, min_tri_angle, point3D_ids):
        ids = list(point3D_ids)
        return self._filter_reprojection(max_reproj_error, ids) + self._filter_small_angles(min_tri_angle, ids)

    def filter_all_points3D(self, max_reproj_error, min_tri_angle):
        # first the reprojection errors, so that a bad observation cannot make a point look stable through a large angle
        n = self._filter_reprojection(max_reproj_error, list(self.mpsfm_rec.points3D.keys()))
        return n + self._filter_small_angles(min_tri_angle, list(self.mpsfm_rec.points3D.keys()))

    def filter_observations_with_negative_depth(self):
        """Deletes every observation of a registered image whose point is not in front of the camera."""
        rec, obs = self.mpsfm_rec, self._obs
        reg = rec.registered_images
        ids, tr = self._tracks(list(rec.points3D.keys()))
        if not ids:
            return 0
        _, _, front = self._numerics(tr, rec.point3D_coordinates(ids), self.device)
        num = 0
        for e in np.flatnonzero(~front):
            image_id, point2D_idx = tr.elements[int(e)]
            if image_id not in reg:
                continue
            p2 = rec.images[image_id].points2D[point2D_idx]
            if not p2.has_point3D():  # its point went away with an earlier deletion (track length 2)
                continue
            obs.delete_observation(image_id, point2D_idx)
            num += 1
        return num
