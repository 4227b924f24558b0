#!/bin/bash
# rocprofv3 kernel-trace stats of the default detect bench (no PMC passes): quick per-kernel time split.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; mkdir -p gpurun_out/prof; export TMPDIR=/tmp
export Y2_TUNE_CACHE=/tmp/y2_tune.json
CMD="python $R/bench.py --steps ${STEPS:-5} --warmup 2 --cpu-sample 0 --train-steps ${TRAIN_STEPS:-0} ${BENCH_ARGS}"
$CMD > /dev/null 2>&1   # populate the autotune cache so the profiled run contains only steady-state launches
cd /tmp
rm -rf $R/gpurun_out/prof/trace
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o trace -- $CMD > $R/gpurun_out/prof/trace.log 2>&1
python3 $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/prof/trace -name '*.db' | head -1) > $R/gpurun_out/prof/kernel_stats.txt
grep -E '^\{' $R/gpurun_out/prof/trace.log | tail -1 > $R/gpurun_out/prof/bench_under_trace.json
head -${HEAD:-16} $R/gpurun_out/prof/kernel_stats.txt | cut -c1-160
find $R/gpurun_out/prof -name '*.db' -delete
