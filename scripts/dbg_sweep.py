import sys, numpy as np
sys.path.insert(0,'.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob,_=make_config(sys.argv[1] if len(sys.argv)>1 else "C3")
h=capi.BAHandle(prob)
L=capi.lib()
for f in [0,1,2,4,6,7,8,16,17]:
    L.mpsfm_debug_set(f<<8)
    ts=[h.sweep_once(1e4) for _ in range(8)][3:]
    print("sweep flags",f,"ms %.3f"%np.mean(ts), flush=True)
