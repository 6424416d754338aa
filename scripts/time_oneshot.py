"""One-shot mpsfm_ba_solve latency (what Optimizer.ba() pays per call) for several problem sizes."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config, make_scene

cases = [("tiny", make_config("tiny")[0]), ("local 12 cams / 4k pts", make_scene(12, 4000, True, seed=3)[0]),
         ("C2", make_config("C2")[0]), ("40 cams / 30k pts prior", make_scene(40, 30000, True, seed=4)[0])]
if len(sys.argv) > 1:
    cases.append(("C3", make_config("C3")[0]))
capi.ba_solve(cases[0][1].copy())
for name, base in cases:
    ts = []
    for i in range(5):
        p = base.copy()
        t0 = time.perf_counter(); s = capi.ba_solve(p); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:28s} one-shot min {min(ts):7.2f} ms  median {sorted(ts)[2]:7.2f} ms  iters {s['num_iterations']}", flush=True)
p = cases[1][1].copy()
capi.ba_solve(p, capi.default_options(verbose=2))
