import sys
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config("C3")
h = capi.BAHandle(prob)
print([h.sweep_once(1e4) for _ in range(4)])
