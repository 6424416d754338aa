"""Single-launch solver of small problems against the launch chain: traces, final state, time per iteration."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene

def run(prob, local):
    capi.lib().mpsfm_debug_set((64 << 8) if local else 0)
    os.environ["MPSFM_LOCAL_LM"] = "1" if local else "0"
    h = capi.BAHandle(prob)
    for _ in range(3):
        h.reset_state(); s = h.solve()
    out = prob.copy()
    h.get_state(out)
    import ctypes as C
    clk = (C.c_int64 * 12)()
    if capi.lib().mpsfm_debug_local_clocks(h._h, clk):
        n = max(clk[6], 1)
        print("   phases us/it: sweep %.1f bar1 %.1f dense+cams %.1f update %.1f bar2 %.1f decide %.1f" % tuple(clk[i] / 100.0 / n for i in range(6)))
        print("   dense: assemble %.1f stacked %.1f barrier %.1f trailing %.1f backsub %.1f" % tuple(clk[i] / 100.0 / n for i in range(7, 12)))
    h.close()
    return s, (out.cam_quat.copy(), out.cam_t.copy(), out.pts.copy())

for ncam, npts in ((3, 300), (6, 1500), (12, 4000), (16, 6000)):
    prob, _ = make_scene(ncam, npts, True, seed=3)
    s0, st0 = run(prob, False)
    s1, st1 = run(prob, True)
    it0, it1 = s0["num_iterations"], s1["num_iterations"]
    dq = max(np.abs(np.asarray(a) - np.asarray(b)).max() for a, b in zip(st0, st1))
    print(f"{ncam} cams / {npts} pts: chain {it0} it {1e6*s0['time_total_s']/max(it0,1):.1f} us/it final {s0['final_cost']:.12e} | "
          f"local {it1} it {1e6*s1['time_total_s']/max(it1,1):.1f} us/it final {s1['final_cost']:.12e} term {s0['termination']}/{s1['termination']} "
          f"max state diff {dq:.3e}", flush=True)
    print("   local phases us/it: sweep %.1f dense+cams %.1f update+decide %.1f" % tuple(1e6 * s1[k] / max(it1, 1) for k in ("time_linearize_s", "time_dense_s", "time_update_s")))
    n = min(len(s0["trace_cost"]), len(s1["trace_cost"]))
    rel = np.abs(np.array(s0["trace_cost"][:n]) - np.array(s1["trace_cost"][:n])) / np.abs(np.array(s0["trace_cost"][:n]))
    print("   trace rel diff max %.3e" % rel.max(), flush=True)
