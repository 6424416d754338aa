"""Generates the committed golden vectors in tests/golden/*.npz.

The reference (tauzn-clock/mpsfm) ships no tests or fixtures for its BA path and its native
dependencies (pyceres 2.4 / Ceres 2.1.0 / an unpinned COLMAP fork) cannot be installed offline,
so these vectors come from INDEPENDENT implementations of the same mathematics available in
this image:

  jacobians.npz    residuals and tangent-space Jacobians of the PINHOLE reprojection functor
                   and the log-depth functor from torch.autograd (float64) with the
                   EigenQuaternionManifold plus  q [+] d = exp(d) * q.
  loss_table.npz   rho, rho' of TRIVIAL / SOFT_L1 / CAUCHY from SciPy's own loss
                   implementations (scipy.optimize._lsq.least_squares), mapped to Ceres'
                   convention rho_ceres(s) = a^2 rho_scipy(s / a^2).
  scene_*.npz      small BA problems + the minimum found by scipy.optimize.least_squares on
                   the identical robustified objective 1/2 sum_i m_i rho_i(|f_i|^2) with the same
                   gauge (camera 0 fixed, camera 1 translation-x fixed).
  grid_sample.npz  bilinear sampling of a map at keypoints by torch grid_sample
                   (align_corners=True, zero padding), the operation of reference
                   mpsfm/sfm/scene/image/mixins/priorutils.py:49-62.
  robust_stats.npz median / MAD cases for fit_robust_gaussian_mad
                   (reference bundle_adjustment.py:10-15).

Run:  python tests/golden/make_golden.py      (needs torch + scipy; CPU only)
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch
from scipy.optimize import least_squares
from scipy.optimize._lsq.least_squares import IMPLEMENTED_LOSSES

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from mpsfm_amd.synthetic import make_scene  # noqa: E402

torch.set_default_dtype(torch.float64)


# ---- independent torch model ------------------------------------------------------------
def t_quat_mul(p, q):
    px, py, pz, pw = p.unbind(-1)
    qx, qy, qz, qw = q.unbind(-1)
    return torch.stack(
        [
            pw * qx + px * qw + py * qz - pz * qy,
            pw * qy - px * qz + py * qw + pz * qx,
            pw * qz + px * qy - py * qx + pz * qw,
            pw * qw - px * qx - py * qy - pz * qz,
        ],
        -1,
    )


def t_exp(d):
    """exp(d) = [sin|d| d/|d|, cos|d|]  (Ceres EigenQuaternionManifold, no 1/2 factor)."""
    n = torch.sqrt((d * d).sum(-1, keepdim=True) + 1e-300)
    return torch.cat([torch.sin(n) / n * d, torch.cos(n)], -1)


def t_rotate(q, v):
    """Eigen quaternion * vector: v + 2w (u x v) + 2 u x (u x v)."""
    u, w = q[..., :3], q[..., 3:4]
    uv = 2.0 * torch.linalg.cross(u, v)
    return v + w * uv + torch.linalg.cross(u, uv)


def t_reproj(q, delta, t, K, X, xy):
    qq = t_quat_mul(t_exp(delta), q)
    Xc = t_rotate(qq, X) + t
    return torch.stack([K[..., 0] * Xc[..., 0] / Xc[..., 2] + K[..., 2] - xy[..., 0],
                        K[..., 1] * Xc[..., 1] / Xc[..., 2] + K[..., 3] - xy[..., 1]], -1)


def t_depth(q, delta, t, X, d):
    qq = t_quat_mul(t_exp(delta), q)
    Xc = t_rotate(qq, X) + t
    return (torch.log(Xc[..., 2]) - torch.log(d)).unsqueeze(-1)


def t_rho(kind, a, s):
    if kind == 1:
        return 2 * a * a * (torch.sqrt(1 + s / (a * a)) - 1)
    if kind == 2:
        return a * a * torch.log(1 + s / (a * a))
    return s


# ---- 1. Jacobians -----------------------------------------------------------------------
def gen_jacobians(n=128, seed=1):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(size=(n, 3))
    X = rng.normal(size=(n, 3))
    # make the point sit in front of the camera: shift translation z so that Zc in [2, 8]
    K = np.stack([rng.uniform(500, 2000, n), rng.uniform(500, 2000, n), rng.uniform(300, 900, n), rng.uniform(300, 900, n)], 1)
    xy = rng.uniform(0, 1600, size=(n, 2))
    d = rng.uniform(1.0, 9.0, n)
    qt, Xt = torch.tensor(q), torch.tensor(X)
    Y = t_rotate(qt, Xt).numpy()
    t[:, 2] = rng.uniform(2.0, 8.0, n) - Y[:, 2]
    r_rp, Jc_rp, Jp_rp, r_dp, Jc_dp, Jp_dp = [], [], [], [], [], []
    for i in range(n):
        qi, ti, Ki, Xi, xyi, di = (torch.tensor(a[i]) for a in (q, t, K, X, xy, d))
        z3 = torch.zeros(3)
        f = lambda dl, tt, xx: t_reproj(qi, dl, tt, Ki, xx, xyi)
        J = torch.autograd.functional.jacobian(f, (z3, ti, Xi))
        r_rp.append(f(z3, ti, Xi).numpy()); Jc_rp.append(torch.cat([J[0], J[1]], 1).numpy()); Jp_rp.append(J[2].numpy())
        g = lambda dl, tt, xx: t_depth(qi, dl, tt, xx, di)
        J = torch.autograd.functional.jacobian(g, (z3, ti, Xi))
        r_dp.append(g(z3, ti, Xi).numpy()); Jc_dp.append(torch.cat([J[0], J[1]], 1).numpy()); Jp_dp.append(J[2].numpy())
    np.savez_compressed(
        os.path.join(HERE, "jacobians.npz"), q=q, t=t, K=K, X=X, xy=xy, d=d,
        r_reproj=np.array(r_rp), Jc_reproj=np.array(Jc_rp), Jp_reproj=np.array(Jp_rp),
        r_depth=np.array(r_dp), Jc_depth=np.array(Jc_dp), Jp_depth=np.array(Jp_dp),
    )


# ---- 2. loss table ------------------------------------------------------------------------
def gen_losses():
    s = np.concatenate([[0.0], np.logspace(-8, 6, 57)])
    a_vals = np.array([0.05, 0.5, 1.5, 3.0, 40.0])
    out = {"s": s, "a": a_vals}
    for name, sp in (("soft_l1", "soft_l1"), ("cauchy", "cauchy")):
        rho0 = np.zeros((len(a_vals), len(s))); rho1 = np.zeros_like(rho0)
        for i, a in enumerate(a_vals):
            z = s / a**2
            rho = np.empty((3, len(s)))
            IMPLEMENTED_LOSSES[sp](z, rho, cost_only=False)
            rho0[i] = a**2 * rho[0]
            rho1[i] = rho[1]
        out[f"{name}_rho0"], out[f"{name}_rho1"] = rho0, rho1
    np.savez_compressed(os.path.join(HERE, "loss_table.npz"), **out)


# ---- 3. small scenes solved by SciPy ------------------------------------------------------
def scipy_minimum(prob):
    """Minimise 1/2 sum m rho(|f|^2) with cameras parameterised as exp(w) * q0."""
    nc, npt = prob.n_cams, prob.n_pts
    q0 = torch.tensor(prob.cam_quat); t0 = torch.tensor(prob.cam_t); X0 = torch.tensor(prob.pts)
    K = torch.tensor(prob.cam_intr[prob.cam_intr_idx])
    oc, op = torch.tensor(prob.obs_cam, dtype=torch.long), torch.tensor(prob.obs_pt, dtype=torch.long)
    xy = torch.tensor(prob.obs_xy)
    dc, dp = torch.tensor(prob.dobs_cam, dtype=torch.long), torch.tensor(prob.dobs_pt, dtype=torch.long)
    dd, dm, da = torch.tensor(prob.dobs_depth), torch.tensor(prob.dobs_magnitude), torch.tensor(prob.dobs_param)
    cam_free = np.ones((nc, 6), bool)
    cam_free[prob.pose_const.astype(bool)] = False
    if prob.gauge_axis_cam >= 0:
        cam_free[prob.gauge_axis_cam, 3] = False
    pt_free = ~prob.pt_const.astype(bool)
    cidx = np.flatnonzero(cam_free.ravel())
    pidx = np.flatnonzero(np.repeat(pt_free, 3))
    ncf = len(cidx)

    def unpack(x):
        cam = torch.zeros(nc * 6).index_add(0, torch.tensor(cidx), x[:ncf]).reshape(nc, 6)
        dX = torch.zeros(npt * 3).index_add(0, torch.tensor(pidx), x[ncf:])
        return cam[:, :3], t0 + cam[:, 3:], X0 + dX.reshape(npt, 3)

    def fun_t(x):
        w, t, X = unpack(x)
        # smooth robustified residual e = sqrt(m rho(s)/s) f  (|e|^2 = m rho(s); rho(s)/s is
        # smooth at s = 0), so a plain linear-loss least squares minimises the same objective
        r = t_reproj(q0[oc], w[oc], t[oc], K[oc], X[op], xy)
        s = (r * r).sum(-1) + 1e-30
        g = torch.sqrt(prob.reproj_loss_magnitude * t_rho(prob.reproj_loss_type, prob.reproj_loss_scale, s) / s)
        e1 = (g[:, None] * r).reshape(-1)
        if len(dd):
            rd = t_depth(q0[dc], w[dc], t[dc], X[dp], dd)[..., 0]
            sd = rd * rd + 1e-30
            e2 = torch.sqrt(dm * t_rho(prob.depth_loss_type, da, sd) / sd) * rd
            return torch.cat([e1, e2])
        return e1

    def fun(x):
        return fun_t(torch.tensor(x)).numpy()

    def jac(x):
        return torch.func.jacfwd(fun_t)(torch.tensor(x)).numpy()

    x0 = np.zeros(ncf + len(pidx))
    res = least_squares(fun, x0, jac=jac, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-12, max_nfev=300)
    assert res.status > 0, res.message
    w, t, X = unpack(torch.tensor(res.x))
    q = t_quat_mul(t_exp(w), q0)
    return res.cost, q.numpy(), t.numpy(), X.numpy(), res


def gen_scenes():
    specs = {
        "scene_2x20": dict(n_cams=2, n_pts=20, with_depth=True, seed=11),
        "scene_5x200": dict(n_cams=5, n_pts=200, with_depth=True, seed=12),
        "scene_4x120_reproj": dict(n_cams=4, n_pts=120, with_depth=False, seed=13),
    }
    for name, kw in specs.items():
        prob, _ = make_scene(**kw)
        cost, q, t, X, res = scipy_minimum(prob)
        print(f"{name}: scipy cost {cost:.12g} nfev {res.nfev} status {res.status} optimality {res.optimality:.3g}")
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            cam_quat=prob.cam_quat, cam_t=prob.cam_t, pts=prob.pts, cam_intr=prob.cam_intr,
            cam_intr_idx=prob.cam_intr_idx, pose_const=prob.pose_const, pt_const=prob.pt_const,
            obs_cam=prob.obs_cam, obs_pt=prob.obs_pt, obs_xy=prob.obs_xy,
            gauge_axis_cam=prob.gauge_axis_cam, reproj_loss_type=prob.reproj_loss_type,
            reproj_loss_scale=prob.reproj_loss_scale, reproj_loss_magnitude=prob.reproj_loss_magnitude,
            dobs_cam=prob.dobs_cam, dobs_pt=prob.dobs_pt, dobs_depth=prob.dobs_depth,
            dobs_magnitude=prob.dobs_magnitude, dobs_param=prob.dobs_param, depth_loss_type=prob.depth_loss_type,
            scipy_cost=cost, scipy_cam_quat=q, scipy_cam_t=t, scipy_pts=X,
        )


# ---- 4. bilinear sampling -------------------------------------------------------------------
def gen_grid_sample(seed=3):
    rng = np.random.default_rng(seed)
    H, W = 29, 39
    data = rng.uniform(0.5, 9.0, (H, W))
    mask = (rng.uniform(size=(H, W)) > 0.15).astype(np.float64)
    sx, sy = (W) / 1600.0, (H) / 1200.0  # camera.sx / camera.sy style scale factors
    kps = np.concatenate([rng.uniform(-40, 1640, (300, 2)), np.array([[0, 0], [1599.9, 1199.9], [(W - 1) / sx, (H - 1) / sy]])])
    kp = torch.tensor(kps * np.array([sx, sy]))
    g = kp.clone()
    g[:, 0] = g[:, 0] / (W - 1) * 2 - 1
    g[:, 1] = g[:, 1] / (H - 1) * 2 - 1
    g = g[None, None]
    out = {}
    for nm, arr in (("data", data), ("mask", mask)):
        smp = torch.nn.functional.grid_sample(torch.tensor(arr)[None, None], g, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0, 0]
        out["sampled_" + nm] = smp.numpy()
    np.savez_compressed(os.path.join(HERE, "grid_sample.npz"), data=data, mask=mask, kps=kps, sx=sx, sy=sy, **out)


# ---- 5. robust statistics ------------------------------------------------------------------
def gen_robust(seed=4):
    rng = np.random.default_rng(seed)
    cases = [rng.normal(0.3, 2.0, 1001), np.concatenate([rng.normal(0, 1, 500), rng.uniform(-50, 50, 60)]), rng.standard_cauchy(400)]
    out = {}
    for i, c in enumerate(cases):
        mu = np.median(c)
        out[f"x{i}"] = c
        out[f"mu{i}"] = mu
        out[f"sigma{i}"] = 1.4826 * np.median(np.abs(c - mu))
    np.savez_compressed(os.path.join(HERE, "robust_stats.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["jacobians", "losses", "grid_sample", "robust", "scenes"]
    for w in which:
        globals()["gen_" + w]()
