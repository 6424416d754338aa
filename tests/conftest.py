import os
import sys

import numpy as np
import pytest

# The CPU oracle is OpenMP code.  On a shared host (the GPU boxes show 256 CPUs and load averages of 50+) 256 spinning threads turn
# a 2 s solve into minutes: a bounded team that sleeps at its barriers, unless the caller says otherwise.  Set before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_scene(name):
    """Golden scene fixture -> (BAProblem, npz)."""
    from mpsfm_amd.problem import BAProblem

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    prob = BAProblem(
        cam_quat=z["cam_quat"], cam_t=z["cam_t"], pts=z["pts"], cam_intr=z["cam_intr"],
        cam_intr_idx=z["cam_intr_idx"], pose_const=z["pose_const"], pt_const=z["pt_const"],
        obs_cam=z["obs_cam"], obs_pt=z["obs_pt"], obs_xy=z["obs_xy"], gauge_axis_cam=int(z["gauge_axis_cam"]),
        reproj_loss_type=int(z["reproj_loss_type"]), reproj_loss_scale=float(z["reproj_loss_scale"]),
        reproj_loss_magnitude=float(z["reproj_loss_magnitude"]), dobs_cam=z["dobs_cam"], dobs_pt=z["dobs_pt"],
        dobs_depth=z["dobs_depth"], dobs_magnitude=z["dobs_magnitude"], dobs_param=z["dobs_param"],
        depth_loss_type=int(z["depth_loss_type"]),
    )
    return prob, z


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
