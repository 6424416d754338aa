#!/bin/bash
# Kernel-variant sweep on the GPU box; build.py rebuilds every object when MPSFM_EXTRA_FLAGS changes.
run() { echo "== $1"; MPSFM_EXTRA_FLAGS="$1" python mpsfm_amd/build.py > /dev/null 2>&1 || { echo build failed; return; }; MPSFM_EXTRA_FLAGS="$1" "${@:2}"; }
for OM in 192 256; do run "-DMPSFM_OBS_MAX=$OM" bash -c 'python scripts/dbg_sweep.py 2>&1 | grep -E "flags (0|4) "'; done
for IP in 8 16 32; do run "-DMPSFM_ITEM_PAIRS=$IP" bash -c 'python scripts/dbg_sweep.py 2>&1 | grep -E "flags (0|4) "'; done
run "-DMPSFM_ENT_STAGE=2048" bash -c 'python scripts/dbg_sweep.py 2>&1 | grep -E "flags (0|4) "'
for PIX in 1 4; do run "-DMPSFM_INT_PIX=$PIX" bash -c 'python scripts/bench_integration.py 2>&1 | grep "^hip" | tail -1'; done
python mpsfm_amd/build.py > /dev/null 2>&1
