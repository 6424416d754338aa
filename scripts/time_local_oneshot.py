"""Local-BA sized one-shot mpsfm_ba_solve: where the time goes, device table build against the host build."""
import os, sys, time
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
base = make_scene(12, 4000, True, seed=3)[0]
capi.ba_solve(base.copy())
for dev in ("1", "0"):
    os.environ["MPSFM_DEV_BUILD"] = dev
    ts = []
    for i in range(7):
        p = base.copy(); t0 = time.perf_counter(); s = capi.ba_solve(p); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"MPSFM_DEV_BUILD={dev}: one-shot min {min(ts):.2f} median {sorted(ts)[3]:.2f} ms, {s['num_iterations']} iterations, solve {1e3*s['time_total_s']:.2f} ms", flush=True)
    capi.ba_solve(base.copy(), capi.default_options(verbose=2))
