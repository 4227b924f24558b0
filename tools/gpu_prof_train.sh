#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; mkdir -p gpurun_out/prof_train; export TMPDIR=/tmp
export Y2_TUNE_CACHE=/tmp/y2_tune_train.json
python $R/bench.py --no-detect --no-conv3 --cpu-sample 0 --train-steps 2 > /dev/null 2>&1
cd /tmp
rm -rf $R/gpurun_out/prof_train/*
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train/trace -o trace -- python $R/bench.py --no-detect --no-conv3 --cpu-sample 0 --train-steps ${TRAIN_STEPS:-4} > $R/gpurun_out/prof_train/trace.log 2>&1
python3 $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/prof_train/trace -name '*.db' | head -1) > $R/gpurun_out/prof_train/kernel_stats.txt
grep -E '^\{' $R/gpurun_out/prof_train/trace.log | tail -1 > $R/gpurun_out/prof_train/bench_under_trace.json
cut -c1-80,100- $R/gpurun_out/prof_train/kernel_stats.txt | head -30
python3 $R/tools/rocprof_summary.py list $(find $R/gpurun_out/prof_train/trace -name '*.db' | head -1) ${LIST_FRAC:-0.80} 420 > $R/gpurun_out/prof_train/dispatch_list.txt
find $R/gpurun_out/prof_train -name '*.db' -delete
