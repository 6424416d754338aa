import os, sys
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
base = make_scene(12, 4000, True, seed=3)[0]
for _ in range(3): capi.ba_solve(base.copy())
os.environ["MPSFM_DEVBUILD_TRACE"] = "1"
h = capi.BAHandle(base.copy(), options=capi.default_options(verbose=2)); h.close()
