#!/bin/bash
# Kernel-variant sweep on the GPU box; build.py rebuilds every object when MPSFM_EXTRA_FLAGS changes.
run() { echo "== $1"; MPSFM_EXTRA_FLAGS="$1" python mpsfm_amd/build.py > /dev/null 2>&1 || { echo build failed; return; }; MPSFM_EXTRA_FLAGS="$1" "${@:2}"; }
for V in ${VARIANTS:-"-DMPSFM_ITEM_PAIRS=16" "-DMPSFM_ITEM_PAIRS=32" "-DMPSFM_ITEM_PAIRS=64" "-DMPSFM_ITEM_PAIRS=128"}; do
  run "$V" bash -c 'python scripts/dbg_sweep.py 2>&1 | grep -E "flags (0|4) "'
done
python mpsfm_amd/build.py > /dev/null 2>&1
