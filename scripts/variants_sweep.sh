#!/bin/bash
# round-1 sweep kernel: records per chunk (LDS per workgroup, workgroups per CU)
export MPSFM_SWEEP_GROUPED=0
for om in 128 192 256; do for ip in 64; do
  export MPSFM_EXTRA_FLAGS="-DMPSFM_OBS_MAX=$om -DMPSFM_ITEM_PAIRS=$ip"
  python -m mpsfm_amd.build > /dev/null 2>&1
  echo "obs_max $om item_pairs $ip: $(python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob,_=make_config("C3")
h=capi.BAHandle(prob)
ts=[h.sweep_once(1e4) for _ in range(12)][2:]
s=h.solve()
print("sweep %.4f ms; solve %.2f ms (sweep %.2f update %.2f)" % (np.mean(ts), 1e3*s["time_total_s"], 1e3*s["time_linearize_s"], 1e3*s["time_update_s"]))
PY
)"
done; done
