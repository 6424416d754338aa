"""Times the depth-from-normals integration (row f1) at the reference's map size: HIP vs the SciPy oracle."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic_maps import make_maps
from oracle import integration_oracle as IO

maps = make_maps(290, 387, seed=8, n_sparse=1500)
nu = maps["normals_uncertainty"]
nvar = np.stack([nu[..., 0, 0], nu[..., 1, 1], nu[..., 2, 2]], -1)
args = (maps["depth_prior"], maps["depth_uncertainty"], maps["valid"], maps["normals"], nvar, maps["depth_init"], maps["K"],
        maps["kps"], maps["depth3d"], maps["zvars3d"])
for i in range(3):
    t0 = time.perf_counter(); d, s, *_ = capi.integrate_depth(*args); t1 = time.perf_counter()
    print(f"hip: wall {1e3*(t1-t0):.1f} ms, device {s['ms']:.2f} ms, irls {s['irls_iterations']}, cg {s['cg_iters']}", flush=True)
keys = ("depth_prior", "depth_uncertainty", "valid", "normals", "normals_uncertainty", "depth_init", "K", "kps", "depth3d", "zvars3d")
t0 = time.perf_counter(); do, ch, st, info = IO.integrate(IO.IntInputs(**{k: maps[k] for k in keys})); t1 = time.perf_counter()
print(f"oracle (scipy cg): {1e3*(t1-t0):.1f} ms, cg {info['cg_iters']}; max rel diff {np.max(np.abs(d/do-1)):.2e}")
# algorithmic bytes of one CG iteration: 5-point stencil on N doubles: read d,cr,cd (3), zz,p (2+halo), write p,q (2); update: read p,q,r,z,minv (5) write z,r,zz (3)
N = 290 * 387
print("bytes per CG iteration (algorithmic):", 15 * 8 * N, "-> GB/s at measured rate:", 15 * 8 * N * sum(s['cg_iters']) / (s['ms'] * 1e-3) / 1e9)

if len(sys.argv) > 1 and sys.argv[1] == "variances":  # row f4: one solve H y = 1 serves every query pixel
    q = np.stack([np.arange(387), np.arange(387) % 290], 1)
    for i in range(2):
        t0 = time.perf_counter(); v, sv = capi.integration_variances(*args[:7], q); t1 = time.perf_counter()
        print(f"variances: wall {1e3*(t1-t0):.1f} ms, device {sv['ms']:.2f} ms, cg {sv['cg_iterations']}, converged {sv['converged']}", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "batch":  # 12 full-size maps: one by one vs one batch
    cases = [make_maps(290, 387, seed=300 + i, n_sparse=1500) for i in range(12)]
    def item(m):
        nu = m["normals_uncertainty"]
        return dict(depth_prior=m["depth_prior"], depth_uncertainty=m["depth_uncertainty"], valid=m["valid"], normals=m["normals"],
                    normals_var=np.stack([nu[..., 0, 0], nu[..., 1, 1], nu[..., 2, 2]], -1), depth_init=m["depth_init"], K=m["K"],
                    kps=m["kps"], depth3d=m["depth3d"], zvars3d=m["zvars3d"])
    items = [item(m) for m in cases]
    capi.integrate_depth_batch(items[:2])
    t0 = time.perf_counter(); seq = [capi.integrate_depth_batch([it])[0] for it in items]; t1 = time.perf_counter()
    bat = capi.integrate_depth_batch(items); t2 = time.perf_counter()
    diff = max(float(np.max(np.abs(a[0] / b[0] - 1))) for a, b in zip(seq, bat))
    its = all(a[1]["cg_iters"] == b[1]["cg_iters"] for a, b in zip(seq, bat))
    print(f"12 maps 290x387: one by one {1e3*(t1-t0):.1f} ms, one batch {1e3*(t2-t1):.1f} ms (device {bat[0][1]['ms']:.1f} ms), "
          f"max rel diff {diff:.1e}, same CG counts {its}")
