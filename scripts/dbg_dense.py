import sys, ctypes, numpy as np
sys.path.insert(0,'.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob,_=make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
h=capi.BAHandle(prob)
h.sweep_once(1e4)
L=capi.lib()
# 1: no potrf, 2: no trsm, 4: no MFMA updates, 8: inverse roles dispatched but idle, 16: every workgroup returns at once
for f in [0,8,1,2,4,7,15,16]:
    L.mpsfm_debug_set(f)
    ts=[h.dense_solve_once() for _ in range(8)][3:]
    print("flags",f,"dense ms %.3f"%np.mean(ts), flush=True)
L.mpsfm_debug_set(0)
