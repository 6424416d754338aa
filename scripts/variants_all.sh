#!/bin/bash
# Kernel-variant sweep on the GPU box; build.py rebuilds every object when MPSFM_EXTRA_FLAGS changes.
run() { echo "== $1"; MPSFM_EXTRA_FLAGS="$1" python mpsfm_amd/build.py > /dev/null 2>&1 || { echo build failed; return; }; MPSFM_EXTRA_FLAGS="$1" "${@:2}"; }
for V in ${VARIANTS:-"-DMPSFM_INV_ROWS=1" "-DMPSFM_INV_ROWS=2" "-DMPSFM_INV_ROWS=4" "-DMPSFM_INV_ROWS=8"}; do
  run "$V" bash -c 'python scripts/dbg_dense.py 2>&1 | grep -E "flags (0|8) "'
done
python mpsfm_amd/build.py > /dev/null 2>&1
