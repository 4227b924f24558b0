#!/bin/bash
# round-5 GPU call: forked capture (weight gradients on a side branch of the graph) A/B + parity under it
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/ff; mkdir -p $O
export Y2_TUNE_CACHE=/tmp/y2_tune_ab.json
timeout 300 python tools/train_steady.py 6 6 > /dev/null 2>&1
for rep in 1 2; do
  for fk in 1 0; do
    echo -n "fork=$fk: " >> $O/train_ab.log; Y2_GRAPH_FORK=$fk timeout 300 python tools/train_steady.py 40 8 2>/dev/null | tail -1 >> $O/train_ab.log
  done
done
cat $O/train_ab.log
unset Y2_TUNE_CACHE
timeout 1500 python -m pytest tests/test_gpu_b64.py tests/test_gpu_plan.py -x -q -s > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -v "^WARNING\|amdgpu.ids" $O/tests.log | grep -i "passed\|failed\|error\|worst\|operand forms\|full-width\|rc=" | tail -30
