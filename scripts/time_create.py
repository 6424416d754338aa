"""Where mpsfm_ba_create spends its time (host table build, uploads) for a configuration: python scripts/time_create.py C3"""
import sys, time
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
for i in range(3):
    t0 = time.perf_counter(); h = capi.BAHandle(prob); t1 = time.perf_counter(); h.close()
    print("create %.2f ms" % (1e3 * (t1 - t0)), flush=True)
o = capi.default_options(verbose=2)
h = capi.BAHandle(prob, options=o); h.close()
for i in range(3):
    p = prob.copy(); t0 = time.perf_counter(); s = capi.ba_solve(p); t1 = time.perf_counter()
    print("one-shot %.2f ms (%d iterations, solve %.2f ms)" % (1e3 * (t1 - t0), s["num_iterations"], 1e3 * s["time_total_s"]), flush=True)
capi.ba_solve(prob.copy(), o)
