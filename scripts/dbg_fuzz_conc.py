import sys, faulthandler
faulthandler.dump_traceback_later(200, exit=True)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from concurrent.futures import ThreadPoolExecutor
from test_gpu_fuzz import _random_problem
from mpsfm_amd import capi
from oracle import cpu_oracle as O
seeds = list(range(16, 40))
ref = {}
for s in seeds:
    ref[s] = O.solve(_random_problem(s))
print('oracle done', flush=True)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    probs = {s: _random_problem(s) for s in seeds}
    with ThreadPoolExecutor(max_workers=6) as ex:
        def run(s):
            o = capi.default_options()
            r = capi.ba_solve(probs[s], o)
            print('done', s, flush=True)
            return r
        sums = dict(zip(seeds, ex.map(run, seeds)))
    bad = 0
    for s in seeds:
        sg, so = sums[s], ref[s]
        if abs(sg["final_cost"] - so["final_cost"]) > 1e-6 * so["final_cost"]:
            bad += 1
            print("rep", rep, "seed", s, "it", sg["num_iterations"], so["num_iterations"], sg["termination"], sg["final_cost"], so["final_cost"])
            print("  gpu cost  ", ["%.6e" % x for x in sg["trace_cost"][:10]])
            print("  cpu cost  ", ["%.6e" % x for x in so["trace_cost"][:10]])
            print("  gpu radius", ["%.3e" % x for x in sg["trace_radius"][:10]], sg["trace_accepted"][:10])
            print("  cpu radius", ["%.3e" % x for x in so["trace_radius"][:10]], so["trace_accepted"][:10])
    print("rep", rep, "mismatches", bad, flush=True)
