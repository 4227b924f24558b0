#!/bin/bash
# round-5 GPU call: parity tests of the benchmarked path, operand-pruning A/B, batch-1 probe
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/fb; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_b64.py tests/test_gpu_plan.py -x -q -s > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -v "^WARNING\|amdgpu.ids" $O/tests.log | grep -i "passed\|failed\|error\|worst\|oracle batch\|operand forms\|full-width\|rc=" | tail -30
export Y2_TUNE_CACHE=/tmp/y2_tune_ab.json
timeout 300 python tools/train_steady.py 6 > /dev/null 2>&1
for rep in 1 2; do
  for pr in 1 0; do
    echo -n "prune=$pr: " >> $O/train_ab.log; Y2_PRUNE_OPERANDS=$pr timeout 300 python tools/train_steady.py 40 2>/dev/null | tail -1 >> $O/train_ab.log
  done
done
cat $O/train_ab.log
unset Y2_TUNE_CACHE
for cfg in "0 256" "400 256" "400 768" "1400 256"; do
  set -- $cfg
  Y2_SMALL_DIRECT=$1 Y2_SPLIT_SLOTS=$2 timeout 200 python tools/latency_b1.py 1 2 2>> $O/latency.err | sed "s/^/small_direct=$1 /" >> $O/latency.log
done
cut -c1-330 $O/latency.log
