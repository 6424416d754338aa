"""Phase durations inside the dense sweep for a local-BA sized problem (one workgroup per CU): per chunk size class."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
prob, _ = make_scene(int(sys.argv[1]), int(sys.argv[2]), True, seed=3)
h = capi.BAHandle(prob)
L = capi.lib()
L.mpsfm_debug_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
L.mpsfm_debug_table.restype = C.c_int64
for _ in range(3):
    h.sweep_once(1e4)
L.mpsfm_debug_set(128 << 8)
h.sweep_once(1e4)
n = h.sweep_parts()["dense_chunks"]
buf = np.zeros(n * 64, np.int64)
L.mpsfm_debug_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
capi._check(L.mpsfm_debug_read_trace(h._h, buf.ctypes.data, len(buf)))
L.mpsfm_debug_set(0)
nb = L.mpsfm_debug_table(h._h, 0, None, 0)
cb = np.zeros(nb, np.uint8); L.mpsfm_debug_table(h._h, 0, cb.ctypes.data, nb)
chunks = cb.view(np.int32).reshape(-1, 12)
t = buf.reshape(n, 4, 16)[:, :, :10].astype(np.float64)
d = np.diff(t, axis=2).max(1) / 2170.0   # us, slowest wave per phase
names = ["P0", "bar", "P1", "bar", "P2+3a", "bar", "products", "P4", "bar"]
life = (t[:, :, 9].max(1) - t[:, :, 0].min(1)) / 2170.0
print("us per phase (slowest wave), by cameras of the chunk:  " + "  ".join(names) + "   | life")
for nc in sorted(set(chunks[:, 5])):
    m = chunks[:, 5] == nc
    print("%2d cams (%3d chunks, %2d-%2d landmarks): " % (nc, m.sum(), chunks[m, 3].min(), chunks[m, 3].max()) + "  ".join("%5.1f" % v for v in d[m].mean(0)) + "   | %5.1f (max %5.1f)" % (life[m].mean(), life[m].max()))
