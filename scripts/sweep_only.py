"""Ten track sweeps of a configuration and nothing else (counter passes of the sweep kernel)."""
import sys
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
h = capi.BAHandle(prob)
for _ in range(10):
    h.sweep_once(1e4)
