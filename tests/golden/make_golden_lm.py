"""Golden Levenberg-Marquardt trajectories from an INDEPENDENT dense implementation.

The oracle (oracle/ba_oracle.c) eliminates the landmarks (Schur complement) and works block by block; this
script restates the same published algorithm — Ceres 2.1 TrustRegionMinimizer + LevenbergMarquardtStrategy
with default options, Corrector robustification with rho'' <= 0, Jacobi column scaling from the first
Jacobian — on the FULL dense Jacobian obtained by torch forward-mode differentiation of the residual
functors of make_golden.py, with numpy.linalg.solve on the full normal equations.  Nothing is shared with
the oracle or the HIP path beyond the problem files.  Output: tests/golden/lm_trace_<scene>.npz with the
cost / radius / accepted flag of every iteration and the final state.

Run:  python tests/golden/make_golden_lm.py      (torch + numpy, CPU, about a minute)
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

from make_golden import t_depth, t_exp, t_quat_mul, t_reproj, t_rho  # noqa: E402
from mpsfm_amd.problem import BAProblem  # noqa: E402

torch.set_default_dtype(torch.float64)

OPT = dict(max_it=50, ftol=1e-6, gtol=1e-10, ptol=1e-8, radius0=1e4, max_radius=1e16, min_radius=1e-32, min_rel=1e-3,
           min_diag=1e-6, max_diag=1e32, max_invalid=5)


def t_rho1(kind, a, s):
    """rho'(s) of TRIVIAL / SOFT_L1 / CAUCHY (Ceres loss_function.cc)."""
    if kind == 0:
        return torch.ones_like(s)
    if kind == 1:
        return 1.0 / torch.sqrt(1.0 + s / (a * a))
    return 1.0 / (1.0 + s / (a * a))


def load(name) -> BAProblem:
    z = np.load(os.path.join(HERE, name + ".npz"))
    return BAProblem(**{k: z[k] for k in ("cam_quat", "cam_t", "pts", "cam_intr", "cam_intr_idx", "pose_const", "pt_const", "obs_cam",
                                           "obs_pt", "obs_xy", "dobs_cam", "dobs_pt", "dobs_depth", "dobs_magnitude", "dobs_param")},
                     gauge_axis_cam=int(z["gauge_axis_cam"]), reproj_loss_type=int(z["reproj_loss_type"]),
                     reproj_loss_scale=float(z["reproj_loss_scale"]), reproj_loss_magnitude=float(z["reproj_loss_magnitude"]),
                     depth_loss_type=int(z["depth_loss_type"]))


def dense_lm(prob: BAProblem):
    nc, npt = prob.n_cams, prob.n_pts
    K = torch.tensor(prob.cam_intr[prob.cam_intr_idx])
    oc, op = torch.tensor(prob.obs_cam, dtype=torch.long), torch.tensor(prob.obs_pt, dtype=torch.long)
    xy = torch.tensor(prob.obs_xy)
    dc, dp = torch.tensor(prob.dobs_cam, dtype=torch.long), torch.tensor(prob.dobs_pt, dtype=torch.long)
    dd, dm, da = torch.tensor(prob.dobs_depth), torch.tensor(prob.dobs_magnitude), torch.tensor(prob.dobs_param)
    cam_var = ~prob.pose_const.astype(bool)
    cam_free = np.repeat(cam_var[:, None], 6, 1)
    if prob.gauge_axis_cam >= 0:
        cam_free[prob.gauge_axis_cam, 3] = False      # SubsetManifold(3, [0]) on the translation
    pt_var = ~prob.pt_const.astype(bool)
    cidx = torch.tensor(np.flatnonzero(cam_free.ravel()))
    pidx = torch.tensor(np.flatnonzero(np.repeat(pt_var, 3)))
    ncf = len(cidx)

    def residual_blocks(q, t, X):
        r = t_reproj(q[oc], torch.zeros(len(oc), 3), t[oc], K[oc], X[op], xy)
        rd = t_depth(q[dc], torch.zeros(len(dc), 3), t[dc], X[dp], dd)[..., 0] if len(dd) else torch.zeros(0)
        return r, rd

    def cost_of(q, t, X):
        r, rd = residual_blocks(q, t, X)
        c = 0.5 * prob.reproj_loss_magnitude * t_rho(prob.reproj_loss_type, prob.reproj_loss_scale, (r * r).sum(-1)).sum()
        if len(dd):
            c = c + 0.5 * (dm * t_rho(prob.depth_loss_type, da, rd * rd)).sum()
        return float(c)

    def linearize(q, t, X):
        """corrected residual vector and dense tangent-space Jacobian at (q, t, X)"""
        def raw(x):
            cam = torch.zeros(nc * 6).index_add(0, cidx, x[:ncf]).reshape(nc, 6)
            dX = torch.zeros(npt * 3).index_add(0, pidx, x[ncf:]).reshape(npt, 3)
            r = t_reproj(q[oc], cam[oc, :3], t[oc] + cam[oc, 3:], K[oc], X[op] + dX[op], xy)
            out = [r.reshape(-1)]
            if len(dd):
                out.append(t_depth(q[dc], cam[dc, :3], t[dc] + cam[dc, 3:], X[dp] + dX[dp], dd)[..., 0])
            return torch.cat(out)
        x0 = torch.zeros(ncf + len(pidx))
        f = raw(x0)
        J = torch.func.jacfwd(raw)(x0)
        nr = 2 * len(oc)
        s_r = (f[:nr].reshape(-1, 2) ** 2).sum(-1)
        w_r = torch.sqrt(prob.reproj_loss_magnitude * t_rho1(prob.reproj_loss_type, prob.reproj_loss_scale, s_r)).repeat_interleave(2)
        w = w_r
        if len(dd):
            w = torch.cat([w_r, torch.sqrt(dm * t_rho1(prob.depth_loss_type, da, f[nr:] ** 2))])
        return (w * f).numpy(), (w[:, None] * J).numpy()

    def plus(q, t, X, delta):
        cam = np.zeros(nc * 6); cam[cidx.numpy()] = delta[:ncf]; cam = cam.reshape(nc, 6)
        dX = np.zeros(npt * 3); dX[pidx.numpy()] = delta[ncf:]
        q2 = t_quat_mul(t_exp(torch.tensor(cam[:, :3])), q)
        q2 = torch.where(torch.tensor(cam_var)[:, None], q2, q)   # exp(0) * q == q, but keep constant poses bit-exact
        return q2, t + torch.tensor(cam[:, 3:]), X + torch.tensor(dX.reshape(npt, 3))

    def x_norm_of(q, t, X):
        return float(np.sqrt((q[cam_var] ** 2).sum() + (t[cam_var] ** 2).sum() + (X[pt_var] ** 2).sum()))

    def grad_max_norm(q, t, X, g):
        q2, t2, X2 = plus(q, t, X, -g)
        d = [torch.abs(q2 - q)[cam_var].max() if cam_var.any() else 0.0, torch.abs(t2 - t).max(), torch.abs(X2 - X).max()]
        return float(max(float(v) for v in d))

    q, t, X = torch.tensor(prob.cam_quat), torch.tensor(prob.cam_t), torch.tensor(prob.pts)
    r, J = linearize(q, t, X)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))           # Jacobi scaling, fixed after the first Jacobian
    x_cost = cost_of(q, t, X)
    trace = [(x_cost, OPT["radius0"], 1)]
    term = None
    if grad_max_norm(q, t, X, J.T @ r) <= OPT["gtol"]:
        term = "gradient_tolerance"
    x_norm = x_norm_of(q, t, X)
    radius, dec, it, invalid = OPT["radius0"], 2.0, 0, 0
    while term is None:
        if it >= OPT["max_it"]:
            term = "max_iterations"; break
        if radius <= OPT["min_radius"]:
            term = "min_radius"; break
        it += 1
        Js = J * scale
        H = Js.T @ Js
        D = np.clip(np.diag(H), OPT["min_diag"], OPT["max_diag"]) / radius
        try:
            step_s = np.linalg.solve(H + np.diag(D), -(Js.T @ r))
            ok = np.all(np.isfinite(step_s))
        except np.linalg.LinAlgError:
            ok = False
        mcc = 0.0
        if ok:
            delta = step_s * scale
            m = J @ delta
            mcc = float(-m @ (r + 0.5 * m))
        if not (ok and mcc > 0.0):
            invalid += 1
            if invalid >= OPT["max_invalid"]:
                term = "invalid_steps"
            radius /= dec; dec *= 2.0
            trace.append((x_cost, radius, 0))
            continue
        invalid = 0
        q2, t2, X2 = plus(q, t, X, delta)
        cand = cost_of(q2, t2, X2)
        step_norm = float(np.sqrt(((q2 - q)[cam_var] ** 2).sum() + ((t2 - t) ** 2).sum() + ((X2 - X) ** 2).sum()))
        if step_norm <= OPT["ptol"] * (x_norm + OPT["ptol"]):
            term = "parameter_tolerance"; break
        change = x_cost - cand
        if abs(change) <= OPT["ftol"] * x_cost:
            term = "function_tolerance"; break
        rel = change / mcc
        if rel > OPT["min_rel"]:
            q, t, X = q2, t2, X2
            x_norm = x_norm_of(q, t, X)
            r, J = linearize(q, t, X)
            x_cost = cost_of(q, t, X)
            radius = min(OPT["max_radius"], radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            dec = 2.0
            trace.append((x_cost, radius, 1))
            if grad_max_norm(q, t, X, J.T @ r) <= OPT["gtol"]:
                term = "gradient_tolerance"
        else:
            radius /= dec; dec *= 2.0
            trace.append((x_cost, radius, 0))
    tr = np.array(trace)
    return dict(trace_cost=tr[:, 0], trace_radius=tr[:, 1], trace_accepted=tr[:, 2].astype(np.uint8), num_iterations=it, termination=term,
                final_cost=x_cost, cam_quat=q.numpy(), cam_t=t.numpy(), pts=X.numpy())


def dense_point_covs(prob: BAProblem):
    """Covariance of every landmark given everything else: inverse of k * sum_obs Jp^T Jp with Jp = d(reprojection) /
    d(point) by forward-mode differentiation (trivial loss scaled by the magnitude, as the oracle documents)."""
    q, t, X = torch.tensor(prob.cam_quat), torch.tensor(prob.cam_t), torch.tensor(prob.pts)
    K = torch.tensor(prob.cam_intr[prob.cam_intr_idx])
    oc, op = torch.tensor(prob.obs_cam, dtype=torch.long), torch.tensor(prob.obs_pt, dtype=torch.long)
    xy = torch.tensor(prob.obs_xy)

    def one(qc, tc, Kc, Xp, uv):
        return t_reproj(qc[None], torch.zeros(1, 3), tc[None], Kc[None], Xp[None], uv[None])[0]

    Jp = torch.func.vmap(torch.func.jacfwd(one, argnums=3))(q[oc], t[oc], K[oc], X[op], xy)      # [n_obs, 2, 3]
    H = torch.zeros(prob.n_pts, 3, 3).index_add(0, op, prob.reproj_loss_magnitude * Jp.transpose(1, 2) @ Jp)
    return torch.linalg.inv(H).numpy()


if __name__ == "__main__":
    covs = dense_point_covs(load("scene_4x120_reproj"))
    np.savez_compressed(os.path.join(HERE, "point_covs_scene_4x120_reproj.npz"), covs=covs)
    print("point covariances:", covs.shape, "trace range", np.trace(covs, axis1=1, axis2=2).min(), np.trace(covs, axis1=1, axis2=2).max())
    for name in ("scene_2x20", "scene_5x200", "scene_4x120_reproj"):
        out = dense_lm(load(name))
        print(f"{name}: {out['num_iterations']} iterations, {out['termination']}, cost {out['trace_cost'][0]:.10g} -> {out['final_cost']:.10g}")
        np.savez_compressed(os.path.join(HERE, "lm_trace_" + name + ".npz"), **out)
