"""Golden vectors for the depth-from-normals integration path (tests/golden/integration_*.npz).

Two kinds of evidence:
  1. The importable pieces of the reference — mpsfm.sfm.scene.camera.CameraIntData and
     mpsfm.utils.integration (move_*, sigmoid) — are imported from /root/reference and their outputs
     for several map sizes are stored (masks, index tables, sigmoid samples): the oracle's restatement
     must reproduce them exactly.  (mpsfm/sfm/scene/image/integration.py itself needs cv2 and cholespy
     at import time and cannot be imported here.)
  2. A small synthetic image run through the oracle (scipy.sparse.linalg.cg, like the reference's CPU
     branch): inputs, per-IRLS energies, CG iteration counts and the integrated depth map.

Run in the build container:  python tests/golden/make_golden_integration.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)


def gen_reference_tables():
    sys.path.insert(0, "/root/reference")
    from mpsfm.sfm.scene.camera import CameraIntData  # noqa: E402  (reference, importable)
    from mpsfm.utils.integration import move_bottom, move_left, move_right, move_top, sigmoid  # noqa: E402

    out = {}
    for (H, W) in ((3, 4), (5, 5), (7, 3)):
        c = CameraIntData(H, W)
        for name in ("has_left_mask", "has_left_mask_left", "has_right_mask", "has_right_mask_right", "has_bottom_mask",
                     "has_bottom_mask_bottom", "has_top_mask", "has_top_mask_top", "pixel_idx_left_center",
                     "pixel_idx_right_right", "pixel_idx_top_center", "pixel_idx_bottom_bottom",
                     "pixel_idx_left_left_indptr", "pixel_idx_right_center_indptr", "pixel_idx_top_top_indptr",
                     "pixel_idx_bottom_center_indptr"):
            out[f"{H}x{W}_{name}"] = np.asarray(getattr(c, name))
    m = np.arange(12).reshape(3, 4) % 3 == 0
    for nm, fn in (("left", move_left), ("right", move_right), ("top", move_top), ("bottom", move_bottom)):
        out[f"move_{nm}"] = np.asarray(fn(m))
    out["move_in"] = m
    x = np.concatenate([np.linspace(-50, 50, 41), [-1e4, 1e4, 0.0]])
    out["sigmoid_x"] = x
    for k in (1, 0.5, 3):
        out[f"sigmoid_k{k}"] = np.asarray(sigmoid(x, k))
    np.savez_compressed(os.path.join(HERE, "integration_reference_tables.npz"), **out)


def gen_case():
    from mpsfm_amd.synthetic_maps import make_maps
    from oracle import integration_oracle as IO

    maps = make_maps(24, 32, seed=7, n_sparse=25)
    inp = IO.IntInputs(**{k: maps[k] for k in ("depth_prior", "depth_uncertainty", "valid", "normals", "normals_uncertainty",
                                               "depth_init", "K", "kps", "depth3d", "zvars3d")})
    depth, changed, state, info = IO.integrate(inp)
    assert changed
    np.savez_compressed(
        os.path.join(HERE, "integration_case_24x32.npz"), **{k: np.asarray(v) for k, v in maps.items()},
        out_depth=depth, energies=np.array(info["energies"]), cg_iters=np.array(info["cg_iters"]), wu=state.wu, wv=state.wv,
    )
    print("energies", info["energies"], "cg iters", info["cg_iters"])
    err0 = np.median(np.abs(np.log(maps["depth_prior"] / maps["depth_true"])))
    err1 = np.median(np.abs(np.log(depth / maps["depth_true"])))
    print(f"median |log depth error| prior {err0:.4f} -> integrated {err1:.4f}")


if __name__ == "__main__":
    gen_reference_tables()
    gen_case()
