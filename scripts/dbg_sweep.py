"""Phase ablations of the dense track sweep (timing only, results are wrong): A.dbg flags 1 = no camera-side accumulation,
2 = no matrix products, 4 = no slab stores, 8 = stop after P0, 16 = stop after P1."""
import sys, numpy as np
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
h = capi.BAHandle(prob)
L = capi.lib()
for f in [0, 1, 2, 4, 6, 7]:
    L.mpsfm_debug_set(f << 8)
    ts = []
    for _ in range(10):
        h.sweep_once(1e4)
        ts.append(h.sweep_parts()["dense_ms"])
    print("dense sweep flags", f, "ms %.4f" % np.mean(ts[4:]), flush=True)
L.mpsfm_debug_set(0)
