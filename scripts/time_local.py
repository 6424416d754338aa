"""Resident solve of small (local-BA sized) problems: per-iteration split."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
for ncam, npts in ((6, 1500), (7, 1800), (10, 3000), (12, 4000), (24, 10000)):
    prob, _ = make_scene(ncam, npts, True, seed=3)
    h = capi.BAHandle(prob)
    for _ in range(3):
        h.reset_state(); s = h.solve()
    it = s["num_iterations"]
    print(f"{ncam} cams / {npts} pts: {it} iterations, total {1e3*s['time_total_s']:.3f} ms = {1e6*s['time_total_s']/it:.1f} us/it; "
          f"sweep {1e6*s['time_linearize_s']/it:.1f}, dense {1e6*s['time_dense_s']/it:.1f}, update {1e6*s['time_update_s']/it:.1f} us/it; plan {h.dense_plan()}", flush=True)
