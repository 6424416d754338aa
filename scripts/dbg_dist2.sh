#!/bin/bash
# two ranks on one GPU (gloo on device tensors), with and without device-wide syncs around the collectives
for SYNC in ${SYNCS:-1 0}; do
  D=$(mktemp -d)
  for R in 0 1; do
    MPSFM_DIST_SYNC=$SYNC MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$SYNC RANK=$R WORLD_SIZE=2 timeout -k 10 200 python tests/_dist_gpu_worker.py $D 77 2> $D/err$R.log &
  done
  wait
  echo "sync=$SYNC"; grep -h "mpsfm_ba" $D/err0.log | head -40; python -c "
import json,sys
for r in (0,1):
    d=json.load(open('$D/r%d.json'%r)); print(r, d['iters'], d['initial_cost'], d['final_cost']); print(['%.9e'%v for v in d['trace']])
"
done
python -c "
import sys; sys.path.insert(0,'.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_scene
p,_=make_scene(12,6000,True,seed=77); s=capi.ba_solve(p); print('single', s['num_iterations'], s['initial_cost'], s['final_cost']); print(['%.9e'%v for v in s['trace_cost']]); print(s['trace_radius']); print(s['termination'])
"
