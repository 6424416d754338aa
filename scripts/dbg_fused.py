"""Fused factorisation (k_chol_fused) against the per-step path on the same reduced system, timing and the phase
timeline of the chain items (wall_clock64 stamps, 10 ns units)."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
prob, _ = make_config(cfg)
os.environ["MPSFM_CHOL_FUSED"] = "0"
h0 = capi.BAHandle(prob)
os.environ["MPSFM_CHOL_FUSED"] = "1"
h1 = capi.BAHandle(prob)
for h in (h0, h1): h.sweep_once(1e4)
t0 = [h0.dense_solve_once() for _ in range(8)][3:]
t1 = [h1.dense_solve_once() for _ in range(8)][3:]
y0, y1 = h0.dense_solution(), h1.dense_solution()
S, rhs = h0.reduced_system()
print("per-step %.3f ms, fused %.3f ms" % (np.mean(t0), np.mean(t1)))
print("max |y_fused - y_step| / max|y| = %.3e" % (np.abs(y0 - y1).max() / np.abs(y0).max()), " finite:", np.isfinite(y1).all())
L = capi.lib()
nt = (h1.reduced_dim + 31) // 32
buf = torch.zeros((nt + 1) * 2 * 8, dtype=torch.int64, device="cuda")
L.mpsfm_debug_set_chol_trace.argtypes = [C.c_void_p]
assert L.mpsfm_debug_set_chol_trace(buf.data_ptr()) == 0
ms = h1.dense_solve_once()
torch.cuda.synchronize()
L.mpsfm_debug_set_chol_trace(None)
t = buf.cpu().numpy().reshape(nt + 1, 2, 8)[:, 0].astype(np.float64) * 0.01
print("fused, traced solve; chain item of column k: entry | flags seen | tiles in LDS | factor done | flags raised (us since column 0)")
base = t[0, 0]
for k in range(0, nt - 1, 3):
    a = t[k] - base
    print("col %3d  entry %7.2f  seen %7.2f  staged +%.2f  factor +%.2f  publish +%.2f | seen->seen of next %.2f" % (
        k, a[0], a[1], a[2] - a[1], a[3] - a[2], a[4] - a[3], (t[k + 1, 1] - t[k, 1]) if k + 1 < nt - 1 else float('nan')))
d = np.diff(t[:nt - 1, 1])
print("column to column (flags seen): mean %.2f min %.2f max %.2f us; factor mean %.2f; publish mean %.2f; raised->seen mean %.2f" % (
    d.mean(), d.min(), d.max(), (t[:nt - 1, 3] - t[:nt - 1, 2]).mean(), (t[:nt - 1, 4] - t[:nt - 1, 3]).mean(), (t[1:nt - 1, 1] - t[:nt - 2, 4]).mean()))
print("first entry -> last diagonal item done: %.2f us" % (t[nt - 1, 3] - base))
for f in [32, 1, 5]:
    L.mpsfm_debug_set(f)
    ts = [h1.dense_solve_once() for _ in range(6)][2:]
    print("dbg flags", f, "fused dense ms %.3f" % np.mean(ts), flush=True)
L.mpsfm_debug_set(0)
