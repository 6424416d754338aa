#!/bin/bash
# CG kernels with 1 / 2 / 4 / 8 pixels per thread
for PIX in 1 2 4 8; do
  MPSFM_EXTRA_FLAGS="-DMPSFM_INT_PIX=$PIX" python mpsfm_amd/build.py > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "PIX=$PIX"; python scripts/bench_integration.py 2>&1 | grep "^hip" | tail -1
done
python mpsfm_amd/build.py > /dev/null 2>&1
