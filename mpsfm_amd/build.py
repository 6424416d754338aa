"""Builds libmpsfm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpsfm_hip.so")
SOURCES = ["ba_kernels.hip", "sweep_dense.hip", "local_lm.hip", "build_dev.hip", "dense_chol.hip", "ba_solver.hip", "tri_kernels.hip", "int_kernels.hip", "prior_kernels.hip", "triangulator.hip", "chol_plan.hip"]
HEADERS = ["common.h", "devbuild.h", "sweep_common.h", "sweep_dense_body.h", "sweep_update_body.h", "dense_tile.h", "lm_decide.h", "local_lm.h", "tri_math.h", "chol_plan.h", os.path.join("..", "..", "include", "mpsfm_hip.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


FLAGS_STAMP = os.path.join(HERE, "build", "extra_flags.txt")


def _flags_changed() -> bool:
    """True when MPSFM_EXTRA_FLAGS differs from what the existing objects were compiled with."""
    want = os.environ.get("MPSFM_EXTRA_FLAGS", "").strip()
    have = open(FLAGS_STAMP).read().strip() if os.path.exists(FLAGS_STAMP) else ""
    return want != have


def is_stale() -> bool:
    if not os.path.exists(LIB) or _flags_changed():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    force = force or _flags_changed()  # other flags: every object is rebuilt
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=default", "-Wall", "-Wno-unused-function"]
    flags += os.environ.get("MPSFM_EXTRA_FLAGS", "").split()
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        src = os.path.join(CSRC, s)
        if not force and os.path.exists(o) and all(
            os.path.getmtime(o) > os.path.getmtime(os.path.join(CSRC, d)) for d in [s] + HEADERS
        ):
            continue
        cmd = [_hipcc(), *flags, "-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(FLAGS_STAMP, "w") as f:
        f.write(os.environ.get("MPSFM_EXTRA_FLAGS", "").strip())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
