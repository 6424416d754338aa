"""Phase durations inside k_track_sweep_dense from shader-clock stamps (debug flag 128): per wave and chunk the cycles between
start | P0 done | barrier | P1 done | barrier | P2+P3a done | barrier | products + slab stores done | P4 + partial sums | barrier."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
h = capi.BAHandle(prob)
L = capi.lib()
for _ in range(3):
    h.sweep_once(1e4)
L.mpsfm_debug_set(128 << 8)
h.sweep_once(1e4)
n = h.sweep_parts()["dense_chunks"]
buf = np.zeros(n * 64, np.int64)
L.mpsfm_debug_read_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
capi._check(L.mpsfm_debug_read_trace(h._h, buf.ctypes.data, len(buf)))
L.mpsfm_debug_set(0)
t = buf.reshape(n, 4, 16)[:, :, :10].astype(np.float64)
d = np.diff(t, axis=2)
names = ["P0 (clear, header)", "barrier", "P1 (loads, linearise, atomics)", "barrier", "P2+P3a", "barrier", "products + stores", "P4 + wave sums", "barrier"]
print("chunks", n, "; mean cycles per wave (shader clock) and share of the wave's life")
tot = (t[:, :, 9] - t[:, :, 0]).mean()
for k, nm in enumerate(names):
    print(f"  {nm:34s} {d[:, :, k].mean():9.0f}  {100 * d[:, :, k].mean() / tot:5.1f} %   (max over waves of a chunk: {d[:, :, k].max(1).mean():9.0f})")
t2 = buf.reshape(n, 4, 16).astype(np.float64)
ok = (t2[:, :, 10] > t2[:, :, 2]) & (t2[:, :, 10] < t2[:, :, 3]) & (t2[:, :, 11] >= t2[:, :, 10]) & (t2[:, :, 11] <= t2[:, :, 3])  # waves whose lane 0 holds a record
print(f"  inside P1 ({ok.mean() * 100:.0f} % of the waves): start -> record / landmark data landed {np.mean((t2[:, :, 10] - t2[:, :, 2])[ok]):7.0f}, "
      f"camera row + linearisation {np.mean((t2[:, :, 11] - t2[:, :, 10])[ok]):7.0f}, J^T J + LDS atomics {np.mean((t2[:, :, 3] - t2[:, :, 11])[ok]):7.0f}")
wc = t2[:, :, 13] - t2[:, :, 12]
print(f"  shader clock / 100 MHz reference: {np.sum(t2[:, :, 9] - t2[:, :, 0]) / np.sum(wc):.2f} (clock64 ticks per 10 ns); wave life {wc.mean() * 0.01:.2f} us; "
      f"first wave start -> last wave end {(t2[:, :, 13].max() - t2[:, :, 12].min()) * 0.01:.1f} us; sum of wave lives / (1024 SIMDs x 3 slots) = {wc.sum() * 0.01 / 3072:.1f} us")
print(f"  wave life {tot:9.0f} cycles; kernel span {(t[:, :, 9].max() - t[:, :, 0].min()):.0f} cycles")
