#!/bin/bash
# Kernel-variant sweep on the GPU box; build.py rebuilds every object when MPSFM_EXTRA_FLAGS changes.
run() { echo "== $1"; MPSFM_EXTRA_FLAGS="$1" python mpsfm_amd/build.py > /dev/null 2>&1 || { echo build failed; return; }; MPSFM_EXTRA_FLAGS="$1" "${@:2}"; }
for V in ${VARIANTS:-"-DMPSFM_INT_PIX=1" "-DMPSFM_INT_PIX=2" "-DMPSFM_INT_PIX=4"}; do
  run "$V" bash -c 'python scripts/bench_integration.py batch 2>&1 | grep -E "12 maps|^hip" | tail -2'
done
python mpsfm_amd/build.py > /dev/null 2>&1
