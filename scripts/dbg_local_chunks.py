"""Single-launch solver: sweep + flush time of every chunk (last iteration) against its size."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
ncam, npts = int(sys.argv[1]), int(sys.argv[2])
prob, _ = make_scene(ncam, npts, True, seed=3)
L = capi.lib()
L.mpsfm_debug_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
L.mpsfm_debug_table.restype = C.c_int64
L.mpsfm_debug_set(64 << 8)
h = capi.BAHandle(prob)
for _ in range(2):
    h.reset_state(); s = h.solve()
def table(which, dtype):
    n = L.mpsfm_debug_table(h._h, which, None, 0)
    buf = np.zeros(n, np.uint8)
    assert L.mpsfm_debug_table(h._h, which, buf.ctypes.data, n) == n
    return buf.view(dtype)
chunks = table(0, np.int32).reshape(-1, 12)
part = table(26, np.float64).reshape(-1, 4)
us = part[:, 3] / 100.0
print("chunks", len(chunks), "sweep+flush us: min %.1f median %.1f max %.1f" % (us.min(), np.median(us), us.max()))
for nc in sorted(set(chunks[:, 5])):
    m = chunks[:, 5] == nc
    print("  %2d cameras: %3d chunks, %.1f .. %.1f us (records %d..%d, landmarks %d..%d)" % (nc, m.sum(), us[m].min(), us[m].max(), chunks[m, 1].min(), chunks[m, 1].max(), chunks[m, 3].min(), chunks[m, 3].max()))
