cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python scripts/dbg_sweep.py C3 > gpurun_out/r03_dbg_sweep_v2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_a -o a --output-format csv -- python3 scripts/sweep_only.py C3 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_b -o b --output-format csv -- python3 scripts/sweep_only.py C3 > gpurun_out/pmc_b.log 2>&1
for k in k_track_sweep_dense k_reduce_slabs; do echo "== $k"; python scripts/pmc_kernel.py gpurun_out/pmc_a $k; python scripts/pmc_kernel.py gpurun_out/pmc_b $k; done > gpurun_out/r03_sweep_v2_counters.txt 2>&1
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
