import sys, os, importlib.util
sys.path.insert(0, '.'); sys.argv = [sys.argv[0], "0", "0"] + sys.argv[1:]
spec = importlib.util.spec_from_file_location("fz", "scripts/fuzz_local.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import numpy as np
from mpsfm_amd import capi
from oracle import cpu_oracle as O
seed = int(sys.argv[3])
po = fz.problem(seed); so = O.solve(po)
res = {}
for name, env in (("local", None), ("chain", "0")):
    if env is None: os.environ.pop("MPSFM_LOCAL_LM", None)
    else: os.environ["MPSFM_LOCAL_LM"] = env
    pg = fz.problem(seed); sg = capi.ba_solve(pg); res[name] = (sg, pg)
    print(name, sg["termination"], sg["num_iterations"], "%.12e" % sg["final_cost"], "max |dt| vs oracle %.3e" % np.abs(pg.cam_t - po.cam_t).max(), "max |dpts| %.3e" % np.abs(pg.pts - po.pts).max())
print("oracle", so["termination"], so["num_iterations"], "%.12e" % so["final_cost"])
print("local vs chain max |dt| %.3e" % np.abs(res["local"][1].cam_t - res["chain"][1].cam_t).max())
print("n_cams", po.n_cams, "n_pts", po.n_pts, "pose_const", po.pose_const.tolist(), "obs per cam", np.bincount(po.obs_cam, minlength=po.n_cams).tolist())
tr = np.array(so["trace_cost"]); print("oracle trace tail", tr[-5:])
print("local trace tail ", np.array(res["local"][0]["trace_cost"])[-5:])
