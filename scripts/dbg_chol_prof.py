import os, sys
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config("C4")
h = capi.BAHandle(prob)
h.sweep_once(1e4)
ts = [h.dense_solve_once() for _ in range(6)]
print("dense ms", ts)
