#!/bin/bash
# builds a few track-sweep tuning variants on the GPU box and times them
for v in "-DMPSFM_ITEM_PAIRS=8" "-DMPSFM_ITEM_PAIRS=16" "-DMPSFM_ITEM_PAIRS=32" "-DMPSFM_ITEM_PAIRS=16 -DMPSFM_OBS_MAX=256" "-DMPSFM_ITEM_PAIRS=16 -DMPSFM_ENT_STAGE=2048 -DMPSFM_OBS_MAX=256"; do
  echo "=== $v"
  MPSFM_EXTRA_FLAGS="$v" python mpsfm_amd/build.py --force > /tmp/build.log 2>&1 || { echo build failed; tail -3 /tmp/build.log | cut -c1-200; continue; }
  python scripts/dbg_sweep.py 2>&1 | grep -E "flags (0|4|7) "
done
