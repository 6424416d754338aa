"""Track-length sensitivity: sweep time per record for scenes with longer and longer tracks (> 64 cameras per
landmark go through k_long_track_sweep)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
for max_track, lam in [(30, 3.0), (60, 20.0), (100, 60.0), (150, 100.0)]:
    # poisson mean lam -> typical track length 2 + lam
    import mpsfm_amd.synthetic as S
    prob, _ = S.make_scene(200, 20000, True, seed=1, max_track=max_track, track_mean=lam) if "track_mean" in S.make_scene.__code__.co_varnames else (None, None)
    if prob is None:
        print("make_scene has no track_mean"); break
    tl = np.bincount(prob.obs_pt)
    with capi.BAHandle(prob, capi.default_options(verbose=0)) as h:
        ts = [h.sweep_once(1e4) for _ in range(5)][2:]
        s = h.solve()
    print(f"max_track {max_track}: mean track {tl.mean():.1f}, >64: {(tl > 64).mean():.2%}, records {prob.n_obs}, sweep {np.mean(ts):.3f} ms "
          f"({1e6 * np.mean(ts) / prob.n_obs:.1f} ns/record), solve {1e3 * s['time_total_s']:.1f} ms / {s['num_iterations']} it", flush=True)
