import os, sys
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
ncam, npts = int(sys.argv[1]), int(sys.argv[2])
prob, _ = make_scene(ncam, npts, True, seed=3)
for local in ("0", "1"):
    os.environ["MPSFM_LOCAL_LM"] = local
    print("MPSFM_LOCAL_LM=" + local, flush=True)
    s = capi.ba_solve(prob.copy(), capi.default_options(verbose=1, max_num_iterations=6))
    print(s["termination"], s["num_iterations"], ["%.6e" % c for c in s["trace_cost"]], flush=True)
