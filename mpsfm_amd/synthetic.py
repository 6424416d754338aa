"""Seeded synthetic bundle-adjustment scenes.

Cameras on a jittered orbit of radius 10 looking at the centre of a 6x6x4 box of uniform
landmarks; one shared PINHOLE camera (fx=fy=1200, cx=800, cy=600, 1600x1200).  Track length
k = clip(2 + Poisson(3), 2, min(30, Nc)); the k cameras nearest in azimuth that see the point.
Measurements = projection + N(0,1) px noise, 5 % outliers U(-20,20) px, rounded to float16 and
widened to float64 (mirrors reference mpsfm/sfm/scene/correspondences/base.py:123).  Depth
priors follow the reference's variance model (mpsfm/sfm/scene/image/depth.py:15,28,77-103) and
the weights of bundle_adjustment.py:153-161.
"""

from __future__ import annotations

import numpy as np

from .problem import LOSS_CAUCHY, LOSS_SOFT_L1, BAProblem

CONFIGS = {
    # name: (n_cams, n_pts, with_depth)
    "tiny": (6, 300, True),
    "C2": (50, 20_000, False),
    "C3": (200, 150_000, True),
    "C4": (1000, 800_000, True),
    # BASELINE config 5 in shape: 300 images, dense matches (~2 M reprojection observations), priors on
    "C5": (300, 400_000, True),
}

FX = FY = 1200.0
CX, CY = 800.0, 600.0
WIDTH, HEIGHT = 1600.0, 1200.0


def quat_from_R(R: np.ndarray) -> np.ndarray:
    """Rotation matrices [N,3,3] -> unit quaternions [N,4] in (x,y,z,w) order."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    q = np.empty((R.shape[0], 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    for i in range(R.shape[0]):
        m = R[i]
        if tr[i] > 0:
            s = np.sqrt(tr[i] + 1.0) * 2
            q[i] = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q[i] = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q[i] = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q[i] = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def R_from_quat(q: np.ndarray) -> np.ndarray:
    """Unit quaternions [N,4] (x,y,z,w) -> rotation matrices [N,3,3]."""
    q = np.asarray(q, dtype=np.float64).reshape(-1, 4)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def quat_mul(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    px, py, pz, pw = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
    qx, qy, qz, qw = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack(
        [
            pw * qx + px * qw + py * qz - pz * qy,
            pw * qy - px * qz + py * qw + pz * qx,
            pw * qz + px * qy - py * qx + pz * qw,
            pw * qw - px * qx - py * qy - pz * qz,
        ],
        axis=-1,
    )


def _orbit_cameras(rng, n_cams):
    az = np.sort(rng.uniform(0.0, 2 * np.pi, n_cams)) if n_cams > 2 else np.array([0.0, 0.35, 0.7][:n_cams])
    radius = 10.0 + rng.normal(0.0, 0.3, n_cams)
    height = rng.normal(0.0, 0.8, n_cams)
    centers = np.stack([radius * np.cos(az), radius * np.sin(az), height], axis=1)
    target = rng.normal(0.0, 0.2, (n_cams, 3))
    fwd = target - centers
    fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)  # rows: camera x, y, z axes in world
    t = -np.einsum("nij,nj->ni", R, centers)
    return R, t, az


def make_scene(
    n_cams: int,
    n_pts: int,
    with_depth: bool = True,
    seed: int = 0,
    outlier_frac: float = 0.05,
    perturb: bool = True,
    max_track: int = 30,
    shard: int = 0,
    track_mean: float = 3.0,
) -> tuple[BAProblem, dict]:
    """Returns (problem at the perturbed initial state, ground truth dict).

    Cameras (and their perturbation) depend on `seed` only; landmarks, observations and priors
    on (`seed`, `shard`), so ranks of a landmark-sharded run draw different landmarks around the
    same cameras."""
    rng_cam = np.random.default_rng([seed, 0])
    rng = np.random.default_rng([seed, 1, shard])
    R, t, az = _orbit_cameras(rng_cam, n_cams)
    X = rng.uniform([-3.0, -3.0, -2.0], [3.0, 3.0, 2.0], (n_pts, 3))

    kmax = min(max_track, n_cams)
    k = np.clip(2 + rng.poisson(track_mean, n_pts), 2, kmax).astype(np.int64)  # track length 2 + Poisson(track_mean)
    # candidate window: cameras nearest in azimuth to the point's own azimuth
    paz = np.mod(np.arctan2(X[:, 1], X[:, 0]) + rng.normal(0, 0.2, n_pts), 2 * np.pi)
    wmax = int(min(n_cams, 2 * kmax + 8))
    obs_cam_l, obs_pt_l, obs_xy_l, obs_z_l = [], [], [], []
    chunk = 50_000
    for s in range(0, n_pts, chunk):
        e = min(n_pts, s + chunk)
        m = e - s
        # nearest camera in sorted azimuth, then a symmetric window around it
        j0 = np.searchsorted(az, paz[s:e]) % n_cams
        offs = np.zeros(wmax, dtype=np.int64)
        offs[1::2] = np.arange(1, wmax // 2 + 1)[: len(offs[1::2])]
        offs[2::2] = -np.arange(1, (wmax - 1) // 2 + 1)[: len(offs[2::2])]
        cand = (j0[:, None] + offs[None, :]) % n_cams  # [m, wmax] ordered by |offset|
        Xc = np.einsum("mwij,mj->mwi", R[cand], X[s:e]) + t[cand]
        z = Xc[..., 2]
        u = FX * Xc[..., 0] / z + CX
        v = FY * Xc[..., 1] / z + CY
        vis = (z > 0.5) & (u >= 0) & (u < WIDTH) & (v >= 0) & (v < HEIGHT)
        rank = np.cumsum(vis, axis=1)
        take = vis & (rank <= k[s:e, None])
        # points seen by fewer than two cameras: fall back to the two nearest regardless
        few = take.sum(axis=1) < 2
        if few.any():
            take[few] = False
            take[few, :2] = True
        pi, wi = np.nonzero(take)
        obs_cam_l.append(cand[pi, wi].astype(np.int32))
        obs_pt_l.append((pi + s).astype(np.int32))
        obs_xy_l.append(np.stack([u[pi, wi], v[pi, wi]], axis=1))
        obs_z_l.append(z[pi, wi])
        del Xc, z, u, v, vis, rank, take
    obs_cam = np.concatenate(obs_cam_l)
    obs_pt = np.concatenate(obs_pt_l)
    xy_true = np.concatenate(obs_xy_l)
    z_true = np.concatenate(obs_z_l)
    n_obs = obs_cam.shape[0]

    xy = xy_true + rng.normal(0.0, 1.0, (n_obs, 2))
    out = rng.uniform(size=n_obs) < outlier_frac
    xy[out] += rng.uniform(-20.0, 20.0, (int(out.sum()), 2))
    xy = xy.astype(np.float16).astype(np.float64)

    kw = {}
    if with_depth:
        sigma_l = 0.0263
        d = z_true * np.exp(rng.normal(0.0, sigma_l, n_obs))
        gross = rng.uniform(size=n_obs) < 0.03
        d[gross] *= 1.5
        valid = rng.uniform(size=n_obs) >= 0.10
        var = np.maximum((sigma_l * d) ** 2, 0.02**2)
        kw = dict(
            dobs_cam=obs_cam[valid],
            dobs_pt=obs_pt[valid],
            dobs_depth=d[valid],
            dobs_magnitude=(d**2 / np.clip(var, 1e-6, None))[valid],
            dobs_param=(2.0 * np.sqrt(var) / d)[valid],
            depth_loss_type=LOSS_CAUCHY,
        )

    q_true = quat_from_R(R)
    q0, t0, X0 = q_true.copy(), t.copy(), X.copy()
    if perturb:
        axis = rng_cam.normal(size=(n_cams, 3))
        axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        ang = np.deg2rad(0.5)
        dq = np.concatenate([axis * np.sin(ang / 2), np.full((n_cams, 1), np.cos(ang / 2))], axis=1)
        dq[0] = [0, 0, 0, 1]
        q0 = quat_mul(dq, q_true)
        dt = rng_cam.normal(0.0, 0.05, (n_cams, 3))
        dt[0] = 0
        t0 = t + dt
        X0 = X + rng.normal(0.0, 0.05, (n_pts, 3))

    pose_const = np.zeros(n_cams, np.uint8)
    pose_const[0] = 1
    prob = BAProblem(
        cam_quat=q0,
        cam_t=t0,
        pts=X0,
        cam_intr=np.array([[FX, FY, CX, CY]]),
        cam_intr_idx=np.zeros(n_cams, np.int32),
        pose_const=pose_const,
        pt_const=np.zeros(n_pts, np.uint8),
        obs_cam=obs_cam,
        obs_pt=obs_pt,
        obs_xy=xy,
        gauge_axis_cam=1 if n_cams > 1 else -1,
        reproj_loss_type=LOSS_SOFT_L1,
        reproj_loss_scale=1.5,
        reproj_loss_magnitude=1.0,
        **kw,
    )
    truth = {"cam_quat": q_true, "cam_t": t, "pts": X}
    return prob, truth


def make_config(name: str, seed: int = 0, shard: int = 0) -> tuple[BAProblem, dict]:
    n_cams, n_pts, with_depth = CONFIGS[name]
    return make_scene(n_cams, n_pts, with_depth=with_depth, seed=seed, shard=shard)


def algorithmic_bytes_sweep(prob: BAProblem, s_blocks: int | None = None) -> int:
    """Compulsory HBM bytes of one track sweep: 24 B per reprojection block
    (2 indices + xy), 32 B per depth block (2 indices + d, m, a), 24 B per landmark (xyz read),
    56 B per camera (pose read) and the reduced camera system written once: its `s_blocks` 6x6 blocks
    (the block-sparse buffer the solver keeps) plus the three slot vectors; without
    `s_blocks` the dense 8 n^2 of SURVEY.md §8d."""
    n = 6 * prob.n_cams
    s_bytes = 8 * n * n if s_blocks is None else 8 * 36 * int(s_blocks) + 3 * 8 * n
    return 24 * prob.n_obs + 32 * prob.n_dobs + 24 * prob.n_pts + 56 * prob.n_cams + s_bytes


def algorithmic_flops_sweep(prob: BAProblem) -> int:
    """fp64 operations one track sweep has to do (multiply-add = 2): per reprojection block the 2-row functor with its
    Jacobian (~120) and its share of J^T J (U 21, V 6, W 18, g_c 6, g_p 3 entries x 2 rows x 2), per depth block the
    1-row versions, per (camera, landmark) record Z = W F^T and W V^-1 g_p (54 + 36), per landmark the 3x3 factorisation
    (~50) and the Schur products of its camera pairs (upper triangle incl. the diagonal, 6x3 * 3x6 = 216 each)."""
    k = np.bincount(np.unique(np.stack([prob.obs_pt, prob.obs_cam], 1), axis=0)[:, 0], minlength=prob.n_pts).astype(np.int64)
    pairs = int(np.sum(k * (k + 1) // 2))
    per_row = 2 * (21 + 6 + 18 + 6 + 3)
    return int(prob.n_obs * (120 + 2 * per_row) + prob.n_dobs * (60 + per_row) + int(k.sum()) * 90 + prob.n_pts * 50 + pairs * 216)


def local_window(prob: BAProblem, window_cams, ref_cam: int, max_track: int = 15):
    """The flat problem of ONE local bundle adjustment cut out of a global flat problem, as
    ``Optimizer.__build_problem(mode="local")`` assembles it on the reference's objects
    (mpsfm/sfm/mapper/bundle_adjustment.py:85-122, bundle from mapper/base.py:729-749):

      * the window's images are the configuration: first one constant, second one with translation x fixed;
      * every observation of a window image gives a reprojection block (and its depth block, if any);
      * landmarks seen by `ref_cam` with a track shorter than `max_track` are explicitly variable: their observations in
        images outside the window come along with those poses constant;
      * any other landmark whose track is not completely inside the problem is constant.

    Returns (local BAProblem, cam_ids [local camera -> global camera], pt_ids [local landmark -> global landmark])."""
    window_cams = [int(c) for c in window_cams]
    n_cfg = len(window_cams)
    in_cfg = np.zeros(prob.n_cams, bool)
    in_cfg[window_cams] = True
    track_len = np.bincount(prob.obs_pt, minlength=prob.n_pts)
    sel = in_cfg[prob.obs_cam]
    seen_by_ref = np.zeros(prob.n_pts, bool)
    seen_by_ref[prob.obs_pt[prob.obs_cam == ref_cam]] = True
    explicit = seen_by_ref & (track_len < max_track)
    sel_all = sel | explicit[prob.obs_pt]
    cams_extra = np.setdiff1d(np.unique(prob.obs_cam[sel_all]), window_cams)
    cam_ids = np.concatenate([np.array(window_cams, np.int64), cams_extra.astype(np.int64)])
    cam_of = np.full(prob.n_cams, -1, np.int64)
    cam_of[cam_ids] = np.arange(len(cam_ids))
    pt_ids = np.unique(prob.obs_pt[sel_all])
    pt_of = np.full(prob.n_pts, -1, np.int64)
    pt_of[pt_ids] = np.arange(len(pt_ids))
    n_in = np.bincount(prob.obs_pt[sel_all], minlength=prob.n_pts)
    pose_const = np.ones(len(cam_ids), np.uint8)
    pose_const[1:n_cfg] = 0
    dsel = in_cfg[prob.dobs_cam] & (pt_of[prob.dobs_pt] >= 0) if prob.n_dobs else np.zeros(0, bool)
    local = BAProblem(
        cam_quat=prob.cam_quat[cam_ids].copy(), cam_t=prob.cam_t[cam_ids].copy(), pts=prob.pts[pt_ids].copy(),
        cam_intr=prob.cam_intr, cam_intr_idx=prob.cam_intr_idx[cam_ids], pose_const=pose_const,
        pt_const=(track_len[pt_ids] > n_in[pt_ids]).astype(np.uint8),
        obs_cam=cam_of[prob.obs_cam[sel_all]].astype(np.int32), obs_pt=pt_of[prob.obs_pt[sel_all]].astype(np.int32),
        obs_xy=prob.obs_xy[sel_all], gauge_axis_cam=1 if n_cfg > 1 else -1,
        reproj_loss_type=prob.reproj_loss_type, reproj_loss_scale=prob.reproj_loss_scale,
        reproj_loss_magnitude=prob.reproj_loss_magnitude,
        dobs_cam=cam_of[prob.dobs_cam[dsel]].astype(np.int32), dobs_pt=pt_of[prob.dobs_pt[dsel]].astype(np.int32),
        dobs_depth=prob.dobs_depth[dsel], dobs_magnitude=prob.dobs_magnitude[dsel], dobs_param=prob.dobs_param[dsel],
        depth_loss_type=prob.depth_loss_type,
    )
    return local, cam_ids, pt_ids


def shuffle_cameras(prob: BAProblem, seed: int = 0) -> tuple[BAProblem, np.ndarray]:
    """The same problem with its cameras in a random order (an unordered photo collection instead of a sequence).
    Returns (problem, perm) with new camera i = old camera perm[i]; the gauge cameras keep their roles."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(prob.n_cams)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(prob.n_cams)
    q = prob.copy()
    q.cam_quat, q.cam_t = prob.cam_quat[perm].copy(), prob.cam_t[perm].copy()
    q.cam_intr_idx, q.pose_const = prob.cam_intr_idx[perm].copy(), prob.pose_const[perm].copy()
    q.obs_cam = inv[prob.obs_cam].astype(np.int32)
    if prob.n_dobs:
        q.dobs_cam = inv[prob.dobs_cam].astype(np.int32)
    if prob.shift_logscale is not None:
        q.shift_logscale = prob.shift_logscale[perm].copy()
    if prob.gauge_axis_cam >= 0:
        q.gauge_axis_cam = int(inv[prob.gauge_axis_cam])
    return q, perm
