#!/bin/bash
# Round-4 rocprofv3 evidence, detect AND train from ONE invocation (so stats and traffic never drift apart): per leg a kernel-trace
# stats pass, then PMC passes (each in its own run: --pmc only with --kernel-trace).  Summaries land in gpurun_out/prof4/ -> copy to profiles/.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/prof4; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
export Y2_TRAIN_GRAPH=0      # the traced training steps issue the launch sequence eagerly (the same kernels the timed step replays from its hipGraph)
DET="python $R/bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-train --no-direct-leg --no-conv3 --no-split-leg --no-multiscale --no-latency --no-resnet"
TRN="python $R/tools/train_steady.py ${TRAIN_STEPS:-12}"
export Y2_TUNE_CACHE=/tmp/y2_tune_r4.json
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
grep -E '^\{' $O/det_trace.log | tail -1 > $O/detect_b32_bench_under_trace.json
prof trn_trace "" $TRN
python3 $R/tools/rocprof_summary.py stats $(find $O/trn_trace -name '*.db' | head -1) > $O/train_b64_kernel_stats.txt
grep -E '^\{' $O/trn_trace.log | tail -1 > $O/train_b64_steady_under_trace.json
# the same steps with the weight gradients on the MAIN stream: per-kernel durations without co-running kernels - what bench.py's train
# roofline table reports (the two-stream schedule of the timed step inflates the durations of kernels that overlap)
Y2_BWD_STREAMS=1 prof trn_trace1 "" $TRN
python3 $R/tools/rocprof_summary.py stats $(find $O/trn_trace1 -name '*.db' | head -1) > $O/train_b64_single_stream_kernel_stats.txt
python3 $R/tools/rocprof_summary.py by_grid $(find $O/trn_trace1 -name '*.db' | head -1) > $O/train_b64_single_stream_kernel_stats_by_grid.txt
grep -E '^\{' $O/trn_trace1.log | tail -1 > $O/train_b64_single_stream_under_trace.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1)); prof det_pmc$i "$set" $DET; prof trn_pmc$i "$set" $TRN
done
python3 $R/tools/rocprof_summary.py pmc $(find $O/det_pmc* -name '*.db') > $O/detect_b32_pmc_summary.txt
python3 $R/tools/rocprof_summary.py pmc $(find $O/trn_pmc* -name '*.db') > $O/train_b64_pmc_summary.txt
python3 $R/tools/traffic_from_pmc.py $O/detect_b32_pmc_summary.txt > $O/detect_b32_traffic.json
python3 $R/tools/traffic_from_pmc.py $O/train_b64_pmc_summary.txt train > $O/train_b64_traffic.json
find $O -name '*.db' -delete; find $O -type d -empty -delete
head -14 $O/detect_b32_kernel_stats.txt | cut -c1-60,100-; head -16 $O/train_b64_kernel_stats.txt | cut -c1-60,100-
cat $O/train_plain.json $O/train_b64_steady_under_trace.json; grep traffic_bytes $O/*traffic.json
