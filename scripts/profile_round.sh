#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel-trace stats and (in separate passes) the
# FETCH_SIZE / WRITE_SIZE counters of the bench command.  Run on the GPU box from the repo root.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-reps 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/bench_write.json" 2> "$OUT/write.err"
# row f1 / f4 kernels (depth integration, variances) in their own trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_int" -- python3 scripts/bench_integration.py variances > "$OUT/bench_integration.log" 2> "$OUT/trace_int.err"
# MFMA counters (own pass, kernel-trace only): the bench command (C3, n = 1194) and a C4-sized dense solve (n = 5994)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -- $CMD > "$OUT/bench_mfma.json" 2> "$OUT/mfma.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma_c4" -- python3 scripts/dbg_chol_prof.py > "$OUT/mfma_c4.log" 2> "$OUT/mfma_c4.err"
python3 scripts/summarize_profile.py "$OUT" "$TAG"
