"""Run as a subprocess with MPSFM_POISON=1 (tests/test_gpu_more.py): one thread's solve is aborted by a failing
all-reduce hook while kernels are in flight, again and again, while a second thread creates handles and solves on
its own stream.  Blocks freed by the aborted handle must not reach the second thread while still in use: its results
have to equal the oracle's every time."""
import sys
import threading

import numpy as np

from mpsfm_amd import capi
from mpsfm_amd.problem import ALLREDUCE_FN
from mpsfm_amd.synthetic import make_scene
from oracle import cpu_oracle as O


def main():
    prob_a, _ = make_scene(40, 20000, True, seed=3)
    prob_b, _ = make_scene(10, 2000, True, seed=4)
    ref = prob_b.copy()
    s_ref = O.solve(ref)
    stop = threading.Event()
    aborted = [0]
    errors = []

    def failing():
        while not stop.is_set():
            calls = [0]

            def cb(user, buf, count, on_device, stream):
                if on_device and count > 1000:
                    calls[0] += 1
                    if calls[0] == 3:  # the all-reduce of the reduced system in LM iteration 2
                        return -1
                return 0

            opts = capi.default_options()
            opts.allreduce = ALLREDUCE_FN(cb)
            try:
                capi.ba_solve(prob_a.copy(), opts)
                errors.append("the failing hook did not abort the solve")
            except capi.MpsfmHipError as e:
                if e.code != -7:
                    errors.append(f"unexpected error code {e.code}")
                aborted[0] += 1

    t = threading.Thread(target=failing)
    t.start()
    n = 0
    try:
        for _ in range(40):
            p = prob_b.copy()
            s = capi.ba_solve(p)
            n += 1
            if s["num_iterations"] != s_ref["num_iterations"] or abs(s["final_cost"] - s_ref["final_cost"]) > 1e-8 * s_ref["final_cost"]:
                errors.append(f"solve {n}: {s['num_iterations']} it cost {s['final_cost']!r} vs {s_ref['num_iterations']} it {s_ref['final_cost']!r}")
                break
            if not np.allclose(p.pts, ref.pts, atol=1e-6):
                errors.append(f"solve {n}: landmarks differ")
                break
    finally:
        stop.set()
        t.join()
    print(f"aborted solves: {aborted[0]}, clean solves beside them: {n}, errors: {errors}")
    return 1 if errors or aborted[0] == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
