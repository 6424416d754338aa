"""Phase timeline of the diagonal workgroup of every tile column in k_chol_level (wall_clock64 stamps, 10 ns units).
Set MPSFM_CHOL_ND=-1 for one column per launch (the caller's camera order): then consecutive columns are consecutive launches
and the "gap" column is the launch boundary."""
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import torch
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob,_=make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
h=capi.BAHandle(prob)
h.sweep_once(1e4)
L=capi.lib()
nt=h.dense_plan()["tile_columns"]
buf=torch.zeros((nt+1)*2*8, dtype=torch.int64, device="cuda")
L.mpsfm_debug_set_chol_trace.argtypes=[C.c_void_p]
for _ in range(3): h.dense_solve_once()
assert L.mpsfm_debug_set_chol_trace(buf.data_ptr())==0
ms=h.dense_solve_once()
torch.cuda.synchronize()
L.mpsfm_debug_set_chol_trace(None)
t=buf.cpu().numpy().reshape(nt+1,2,8).astype(np.float64)*0.01  # us
print("dense solve %.3f ms, nt %d" % (ms, nt))
w0,w1=t[:nt,0],t[:nt,1]
print("per step (us), wave 0: entry->tiles in LDS | barrier | stacked potrf+trsm | store ; gap to the next kernel's entry")
for j in range(0,nt,4):
    a=w0[j]
    gap = (w0[j+1,0]-a[4]) if j+1<nt else float('nan')
    print("step %2d  load+update %.2f  barrier %.2f  factor %.2f  store %.2f | inside %.2f  gap %.2f" % (j, a[1]-a[0], a[2]-a[1], a[3]-a[2], a[4]-a[3], a[4]-a[0], gap))
d=np.diff(w0[:,0])
print("entry-to-entry: mean %.2f min %.2f max %.2f us" % (d.mean(), d.min(), d.max()))
print("means: load+update %.2f  factor %.2f  store %.2f  inside %.2f  gap %.2f" % ((w0[:,1]-w0[:,0]).mean(), (w0[:,3]-w0[:,2]).mean(), (w0[:,4]-w0[:,3]).mean(), (w0[:,4]-w0[:,0]).mean(), (w0[1:,0]-w0[:-1,4]).mean()))
