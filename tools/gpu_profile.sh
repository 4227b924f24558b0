#!/bin/bash
# rocprofv3 evidence (rounds 5-6), detect AND train from ONE invocation (stats and traffic never drift apart): per leg a kernel-trace stats pass, then
# PMC passes (each in its own run: --pmc only with --kernel-trace).  Summaries land in gpurun_out/prof/ -> copy to profiles/rNN_* (tools/gpu_profile.sh; `PROFILE_ROUND=r06 ... ` copies them itself).
# New in round 5: the training trace is taken on the TIMED launch form (hipGraph replays of the captured step: forked weight gradients, pruned
# operand preparation) - Y2_PROFILE_EAGER=1 falls back to eager launches of the same sequence; the detect record carries the dominant kernel's
# average duration under the timed two-stream schedule (`dominant_trace`).
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
[ "$Y2_PROFILE_EAGER" = "1" ] && export Y2_TRAIN_GRAPH=0
DET="python $R/bench.py --headline detect --steps 8 --warmup 2 --cpu-sample 0 --no-train --no-direct-leg --no-conv3 --no-split-leg --no-multiscale --no-latency --no-resnet --tables $O/det_tables.json"
TRN="python $R/tools/train_steady.py ${TRAIN_STEPS:-16} 6"
export Y2_TUNE_CACHE=/tmp/y2_tune_r5.json
$DET > /dev/null 2>&1; $TRN > $O/train_plain.json 2>/dev/null      # populate the algorithm cache: the profiled runs contain steady-state launches only
cd /tmp
prof() {  # tag, pmc-set-or-empty, command...
  local tag=$1 set=$2; shift 2
  if [ -z "$set" ]; then timeout 900 rocprofv3 --kernel-trace --stats -d $O/$tag -o t -- "$@" > $O/$tag.log 2>&1
  else timeout 900 rocprofv3 --kernel-trace --pmc $set -d $O/$tag -o t -- "$@" > $O/$tag.log 2>&1; fi
}
prof det_trace "" $DET
python3 $R/tools/rocprof_summary.py stats $(find $O/det_trace -name '*.db' | head -1) > $O/detect_b32_kernel_stats.txt
python3 $R/tools/rocprof_summary.py by_grid $(find $O/det_trace -name '*.db' | head -1) > $O/detect_b32_kernel_stats_by_grid.txt
python3 $R/tools/trace_families.py $(find $O/det_trace -name '*.db' | head -1) conv0_kernel 0.3 > $O/detect_b32_trace.json
grep -E '^\{' $O/det_trace.log | tail -1 > $O/detect_b32_bench_under_trace.json
prof trn_trace "" $TRN
python3 $R/tools/rocprof_summary.py stats $(find $O/trn_trace -name '*.db' | head -1) > $O/train_b64_kernel_stats.txt
python3 $R/tools/rocprof_summary.py by_grid $(find $O/trn_trace -name '*.db' | head -1) > $O/train_b64_kernel_stats_by_grid.txt
python3 $R/tools/trace_families.py $(find $O/trn_trace -name '*.db' | head -1) loss_fwd_kernel 0.4 > $O/train_b64_trace.json
grep -E '^\{' $O/trn_trace.log | tail -1 > $O/train_b64_steady_under_trace.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1)); prof det_pmc$i "$set" $DET; prof trn_pmc$i "$set" $TRN
done
python3 $R/tools/rocprof_summary.py pmc $(find $O/det_pmc* -name '*.db') > $O/detect_b32_pmc_summary.txt
python3 $R/tools/rocprof_summary.py pmc $(find $O/trn_pmc* -name '*.db') > $O/train_b64_pmc_summary.txt
python3 $R/tools/traffic_from_pmc.py $O/detect_b32_pmc_summary.txt $O/detect_b32_kernel_stats_by_grid.txt $O/detect_b32_trace.json > $O/detect_b32_traffic.json
python3 $R/tools/traffic_from_pmc.py $O/train_b64_pmc_summary.txt train $O/train_b64_kernel_stats_by_grid.txt $O/train_b64_trace.json > $O/train_b64_traffic.json
STEPS=$(python3 -c "import json;print(json.load(open('$O/train_b64_traffic.json'))['steps_profiled'])")
python3 $R/tools/traffic_by_kernel.py $O/train_b64_pmc_summary.txt $STEPS > $O/train_b64_traffic_by_kernel.txt
STEPS=$(python3 -c "import json;print(json.load(open('$O/detect_b32_traffic.json'))['steps_profiled'])")
python3 $R/tools/traffic_by_kernel.py $O/detect_b32_pmc_summary.txt $STEPS > $O/detect_b32_traffic_by_kernel.txt
find $O -name '*.db' -delete; find $O -type d -empty -delete
head -12 $O/detect_b32_kernel_stats.txt | cut -c1-60,100-; head -14 $O/train_b64_kernel_stats.txt | cut -c1-60,100-
cat $O/train_plain.json $O/train_b64_steady_under_trace.json; grep traffic_bytes $O/*traffic.json; head -3 $O/train_b64_traffic_by_kernel.txt
if [ -n "$PROFILE_ROUND" ]; then
  for f in detect_b32_kernel_stats.txt detect_b32_kernel_stats_by_grid.txt detect_b32_pmc_summary.txt detect_b32_traffic.json detect_b32_traffic_by_kernel.txt detect_b32_bench_under_trace.json \
           train_b64_kernel_stats.txt train_b64_kernel_stats_by_grid.txt train_b64_pmc_summary.txt train_b64_traffic.json train_b64_traffic_by_kernel.txt train_b64_steady_under_trace.json; do
    cp $O/$f $R/gpurun_out/${PROFILE_ROUND}_$f
  done
fi
