"""Depth-block selection of a whole bundle.

``gather_bundle`` collects — through the accessor names the reference itself uses (bundle_adjustment.py:128-144) — the
keypoints that have a 3-D point, their per-keypoint prior variances and 3-D point ids for every image of a bundle and
concatenates them; the per-observation arithmetic of reference bundle_adjustment.py:130-161 / :312-329 then runs in
one HIP launch (``capi.depth_blocks``).  Its NumPy restatement (the checker) is oracle/prior_oracle.py.
"""

from __future__ import annotations

import numpy as np

from ...utils.ids import unique_ids

F_VALID, F_POSITIVE, F_SCALE, F_GROSS = 1, 2, 4, 8


def gather_bundle(rec, imids, depth_type="update"):
    """Concatenated per-observation inputs of the given images (those with an activated depth map and at least one
    observation).  Returns a dict of arrays plus `images` (the image ids, in order) and `point_ids` (sorted unique)."""
    images, maps, valids, sx, sy, quat, trans = [], [], [], [], [], [], []
    o_img, o_xy, o_var, o_pid, o_p2d = [], [], [], [], []
    for imid in imids:
        image = rec.images[imid]
        if not image.depth.activated:
            continue
        p2Ds = np.asarray(image.get_observation_point2D_idxs(), dtype=np.int64)
        if len(p2Ds) == 0:
            continue
        cam = rec.rec.cameras[image.camera_id]
        uu = image.depth.uncertainty_update
        var = np.asarray(uu, np.float64)[p2Ds] if isinstance(uu, np.ndarray) else np.array([uu[int(i)] for i in p2Ds], np.float64)
        k = len(images)
        images.append(imid)
        maps.append(image.depth.data if depth_type == "update" else image.depth.data_prior)
        valids.append(image.depth.valid)
        sx.append(cam.sx); sy.append(cam.sy)
        quat.append(np.asarray(image.cam_from_world.rotation.quat, np.float64)); trans.append(np.asarray(image.cam_from_world.translation, np.float64))
        o_img.append(np.full(len(p2Ds), k, np.int32))
        o_xy.append(np.asarray(image.keypoint_coordinates(p2Ds), np.float64).reshape(-1, 2))
        o_var.append(var)
        o_pid.append(np.asarray(image.point3D_ids(p2Ds), dtype=np.uint64))
        o_p2d.append(p2Ds)
    if not images:
        return None
    pid = np.concatenate(o_pid)
    point_ids, obs_pt = unique_ids(pid)
    return dict(images=images, depth_maps=maps, valid_maps=valids, sx=np.array(sx), sy=np.array(sy), cam_quat=np.array(quat).reshape(-1, 4),
                cam_t=np.array(trans).reshape(-1, 3), obs_img=np.concatenate(o_img), obs_xy=np.concatenate(o_xy), obs_var=np.concatenate(o_var),
                obs_pid=pid, obs_pt=obs_pt.astype(np.int32), obs_p2d=np.concatenate(o_p2d), point_ids=point_ids)
