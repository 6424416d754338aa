import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
base = prob.copy()
for i in range(3):
    p = base.copy()
    t0 = time.perf_counter(); h = capi.BAHandle(p, capi.default_options(verbose=2 if i == 2 else 0)); t1 = time.perf_counter()
    h.options.verbose = 0
    h.close()
    print(f"create {1e3*(t1-t0):.1f} ms", flush=True)
p = base.copy(); t0 = time.perf_counter(); s = capi.ba_solve(p); print(f"one-shot {1e3*(time.perf_counter()-t0):.1f} ms")
p = base.copy(); t0 = time.perf_counter(); s = capi.ba_solve(p, capi.default_options(verbose=2)); print(f"one-shot {1e3*(time.perf_counter()-t0):.1f} ms, iters {s['num_iterations']}")
