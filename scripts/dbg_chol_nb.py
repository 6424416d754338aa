"""Dense solve time at C4 size (n = 5994) for several outer-panel widths (MPSFM_CHOL_NB)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C4")
for nb, ovl in [(10**6, 1), (4, 1), (8, 0), (8, 1), (16, 1)]:
    os.environ["MPSFM_CHOL_NB"] = str(nb)       # read once per handle
    os.environ["MPSFM_CHOL_OVERLAP"] = str(ovl)
    with capi.BAHandle(prob.copy()) as h:
        h.sweep_once(1e4)
        n = h.reduced_dim
        ts = [h.dense_solve_once() for _ in range(5)][2:]
    ms = float(np.mean(ts))
    print(f"NB {nb:>8} overlap {ovl}: dense solve {ms:.3f} ms -> {(n**3 / 3 + 2 * n * n) / (ms * 1e-3) / 1e12:.2f} TFLOP/s", flush=True)
