#!/bin/bash
# Kernel-variant sweep on the GPU box; build.py rebuilds every object when MPSFM_EXTRA_FLAGS changes.
run() { echo "== $1"; MPSFM_EXTRA_FLAGS="$1" python mpsfm_amd/build.py > /dev/null 2>&1 || { echo build failed; return; }; MPSFM_EXTRA_FLAGS="$1" "${@:2}"; }
for V in ${VARIANTS:-"-DMPSFM_LOCAL_CAMS=64" "-DMPSFM_LOCAL_CAMS=128" "-DMPSFM_LOCAL_CAMS=254"}; do
  run "$V" bash -c 'python scripts/dbg_sweep.py 2>&1 | grep -E "flags 0 "; python scripts/dbg_long.py 2>&1 | grep max_track'
done
python mpsfm_amd/build.py > /dev/null 2>&1
