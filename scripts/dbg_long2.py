import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.problem import BAProblem
from mpsfm_amd.synthetic import make_scene, R_from_quat
from oracle import cpu_oracle as O
rng = np.random.default_rng(5)
prob, truth = make_scene(260, 400, True, seed=19)
R = R_from_quat(truth["cam_quat"])
oc, op, oxy, dd, dm, da = [], [], [], [], [], []
for pt in range(3):
    Xc = R @ truth["pts"][pt] + truth["cam_t"]
    uv = np.stack([1200 * Xc[:, 0] / Xc[:, 2] + 800, 1200 * Xc[:, 1] / Xc[:, 2] + 600], 1) + rng.normal(0, 1, (260, 2))
    oc.append(np.arange(260)); op.append(np.full(260, pt)); oxy.append(uv)
    d = Xc[:, 2] * np.exp(rng.normal(0, 0.0263, 260))
    var = np.maximum((0.0263 * d) ** 2, 0.02**2)
    dd.append(d); dm.append(d**2 / var); da.append(2 * np.sqrt(var) / d)
long_prob = BAProblem(prob.cam_quat, prob.cam_t, prob.pts, prob.cam_intr, prob.cam_intr_idx, prob.pose_const, prob.pt_const,
    np.concatenate([prob.obs_cam] + oc).astype(np.int32), np.concatenate([prob.obs_pt] + op).astype(np.int32),
    np.concatenate([prob.obs_xy] + oxy), gauge_axis_cam=prob.gauge_axis_cam,
    dobs_cam=np.concatenate([prob.dobs_cam] + oc).astype(np.int32), dobs_pt=np.concatenate([prob.dobs_pt] + op).astype(np.int32),
    dobs_depth=np.concatenate([prob.dobs_depth] + dd), dobs_magnitude=np.concatenate([prob.dobs_magnitude] + dm),
    dobs_param=np.concatenate([prob.dobs_param] + da), depth_loss_type=prob.depth_loss_type)
for name, pr in (("plain", prob), ("long", long_prob)):
    ref = O.reduced_system(pr, radius=50.0)
    o = capi.default_options(); o.verbose = 2
    with capi.BAHandle(pr.copy(), o) as h:
        h.sweep_once(50.0)
        print(name, h.sweep_parts())
        S, rhs = h.reduced_system()
    D = np.abs(S - ref["S"]) > 1e-9 * np.abs(ref["S"]).max()
    n = S.shape[0] // 6
    blk = D.reshape(n, 6, n, 6).any((1, 3))
    ij = np.argwhere(blk)
    print(name, "bad blocks", len(ij), ij[:20].tolist(), "rhs bad", int((np.abs(rhs - ref["rhs"]) > 1e-9 * np.abs(ref["rhs"]).max()).sum()))
    for i, j in ij[:3]:
        print(" block", i, j, "\n", S[6*i:6*i+6, 6*j:6*j+6], "\n", ref["S"][6*i:6*i+6, 6*j:6*j+6])
