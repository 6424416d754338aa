#!/bin/bash
# bench.py with two ranks on ONE GPU (gloo sums the device buffers): checks the multi-rank code path of the bench,
# weak and strong, without a second card.  Not a measurement.
set -e
for SC in weak strong; do
  PORT=$((29600 + RANDOM % 200))
  HSA_ENABLE_IPC_MODE_LEGACY=0 MPSFM_BENCH_BACKEND=gloo MPSFM_BENCH_DEVICE=0 MPSFM_BENCH_SCALING=$SC timeout -k 10 400 \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT \
    bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline
done
