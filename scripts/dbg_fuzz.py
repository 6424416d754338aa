import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_fuzz import _random_problem
from mpsfm_amd import capi
from oracle import cpu_oracle as O
seeds = [int(x) for x in sys.argv[1:]] or list(range(16, 40))
for s in seeds:
    pg, po = _random_problem(s), _random_problem(s)
    sg, so = capi.ba_solve(pg), O.solve(po)
    flag = "" if abs(sg["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"] else "  <-- MISMATCH"
    print(s, pg.n_cams, pg.n_pts, sg["num_iterations"], so["num_iterations"], sg["termination"], so["termination"], sg["final_cost"], so["final_cost"], flag, flush=True)
    if flag:
        print("  gpu trace", sg["trace_cost"][:12], sg["trace_accepted"][:12])
        print("  cpu trace", so["trace_cost"][:12], so["trace_accepted"][:12])
