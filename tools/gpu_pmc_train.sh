#!/bin/bash
# PMC passes (each in its own rocprofv3 run, --pmc only with --kernel-trace) over a few training steps: where the wave cycles of the
# MFMA kernels of the training step go.  Output: gpurun_out/pmc_train/pmc_summary.txt
cd "$(dirname "$0")/.." || exit 1
R=$PWD; mkdir -p gpurun_out/pmc_train; export TMPDIR=/tmp
export Y2_TUNE_CACHE=/tmp/y2_tune_pmc.json Y2_BWD_STREAMS=1
python $R/tools/train_steps.py 2 > /dev/null 2>&1       # autotune cache
cd /tmp
rm -rf $R/gpurun_out/pmc_train/*
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_train/pmc$i -o pmc -- python $R/tools/train_steps.py 2 > $R/gpurun_out/pmc_train/pmc$i.log 2>&1
done
python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/pmc_train/pmc* -name '*.db') > $R/gpurun_out/pmc_train/pmc_summary.txt
grep -E "conv_wgrad_kernelILi128ELi2ELi128|conv_fwd_dma_kernelILi64ELi128" $R/gpurun_out/pmc_train/pmc_summary.txt | cut -c1-70,95- | head -60
find $R/gpurun_out/pmc_train -name '*.db' -delete
