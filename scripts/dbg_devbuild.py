import sys, time, os
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config, make_scene
for name, prob in (("C3", make_config("C3")[0]), ("local", make_scene(12, 4000, True, seed=3)[0])):
    for i in range(3):
        h = capi.BAHandle(prob); h.close()
    os.environ["MPSFM_DEVBUILD_TRACE"] = "1"
    print("==", name, flush=True)
    h = capi.BAHandle(prob); h.close()
    del os.environ["MPSFM_DEVBUILD_TRACE"]
