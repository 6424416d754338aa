"""Times the parts of the track sweep (dense chunks | slab reduction | general chunks) of a configuration with HIP events."""
import sys
import numpy as np
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
for cfg in (sys.argv[1:] or ["C3"]):
    prob, _ = make_config(cfg)
    h = capi.BAHandle(prob)
    tot, parts = [], []
    for _ in range(12):
        tot.append(h.sweep_once(1e4))
        parts.append(h.sweep_parts())
    p = parts[-1]
    print(cfg, "sweep ms %.4f" % np.mean(tot[4:]), "dense %.4f reduce %.4f general %.4f" % tuple(np.mean([[q["dense_ms"], q["reduce_ms"], q["general_ms"]] for q in parts[4:]], 0)),
          {k: p[k] for k in ("dense_chunks", "general_chunks", "long_tracks", "reduce_parts")}, flush=True)
    h.close()
