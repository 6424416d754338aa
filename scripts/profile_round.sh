#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel-trace stats and (in separate passes, never with a trace domain beside
# --kernel-trace) the FETCH_SIZE / WRITE_SIZE / MFMA / SQ counters of the bench command.  Run on the GPU box from the repo root.
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
DST=gpurun_out/profiles_$TAG
# SQ counters of the track-sweep kernels (two passes), per launch
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/sq_a" --output-format csv -- python3 scripts/sweep_only.py C3 > "$OUT/sq_a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT -d "$OUT/sq_b" --output-format csv -- python3 scripts/sweep_only.py C3 > "$OUT/sq_b.log" 2>&1
{
  echo "# SQ counters of the track-sweep kernels at C3, per launch (scripts/sweep_only.py under rocprofv3 --pmc, two passes; scripts/pmc_kernel.py)."
  echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."
  for k in k_track_sweep_dense k_reduce_slabs; do echo "== $k"; python3 scripts/pmc_kernel.py "$OUT/sq_a" $k; python3 scripts/pmc_kernel.py "$OUT/sq_b" $k; done
} > "$DST/${TAG}_sweep_sq_counters_after.txt" 2>&1
# shader-clock stamps inside the dense sweep, per-launch durations of the factorisation levels, the parts of the sweep, the device build
python3 scripts/dbg_sweep_trace.py C3 > "$DST/${TAG}_sweep_phase_trace.txt" 2>&1
python3 scripts/level_durations.py "$OUT/trace" > "$DST/${TAG}_level_durations.txt" 2>&1
python3 scripts/sweep_parts.py C3 C4 C5 C2 > "$DST/${TAG}_sweep_parts.txt" 2>&1
python3 scripts/dbg_devbuild.py > "$DST/${TAG}_device_build_steps.txt" 2>&1
python3 scripts/time_oneshot.py C3 2>&1 | grep -v "^\[mpsfm_ba\] it " > "$DST/${TAG}_one_shot.txt"
{ python3 scripts/dbg_local.py; python3 scripts/time_local_oneshot.py 2>&1 | grep "one-shot\|create:\|build:"; } 2>&1 | grep -v amdgpu.ids > "$DST/${TAG}_local_lm.txt"
cp "$OUT/bench_trace.json" "$DST/${TAG}_bench_under_rocprof.json"
rm -rf "$OUT/sq_a" "$OUT/sq_b" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/pmc_mfma_c4" "$OUT/trace_int"
