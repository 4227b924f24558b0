#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then PMC passes (each in its own run,
# --pmc only with --kernel-trace; never combined with other trace domains).
cd "$(dirname "$0")/.." || exit 1
R=$PWD; mkdir -p gpurun_out/prof; export TMPDIR=/tmp
export Y2_TUNE_CACHE=/tmp/y2_tune.json
CMD="python $R/bench.py --steps ${STEPS:-8} --warmup 2 --cpu-sample 0 --no-train --no-direct-leg --no-conv3 ${BENCH_ARGS}"
$CMD > /dev/null 2>&1   # populate the tile-autotune cache so the profiled runs contain only steady-state launches
cd /tmp
rm -rf $R/gpurun_out/prof/*
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o trace -- $CMD > $R/gpurun_out/prof/trace.log 2>&1
python3 $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/prof/trace -name '*.db' | head -1) > $R/gpurun_out/prof/kernel_stats.txt
grep -E '^\{' $R/gpurun_out/prof/trace.log | tail -1 > $R/gpurun_out/prof/bench_under_trace.json
head -12 $R/gpurun_out/prof/kernel_stats.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof/pmc$i -o pmc -- $CMD > $R/gpurun_out/prof/pmc$i.log 2>&1
done
python3 $R/tools/rocprof_summary.py pmc $(find $R/gpurun_out/prof/pmc* -name '*.db') > $R/gpurun_out/prof/pmc_summary.txt
python3 $R/tools/traffic_from_pmc.py $R/gpurun_out/prof/pmc_summary.txt > $R/gpurun_out/prof/traffic.json
grep -E "MFMA_BUSY|GRBM_GUI" $R/gpurun_out/prof/pmc_summary.txt | grep -E "conv_fwd|wino" | cut -c1-60,91- | head -30; grep traffic_bytes $R/gpurun_out/prof/traffic.json
find $R/gpurun_out/prof -name '*.db' -delete
