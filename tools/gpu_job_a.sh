#!/bin/bash
# round-5 GPU call: operand-pruning A/B (steady state) with the 4x4-tile operand in y2_prep_weights, parity tests, contention rehearsal
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/fe; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_b64.py tests/test_gpu_plan.py -x -q -s -k "multi_tensor or batch64 or capture_falls or full_width_replays or frozen" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -v "^WARNING\|amdgpu.ids" $O/tests.log | grep -i "passed\|failed\|error\|worst\|oracle batch\|operand forms\|full-width\|rc=" | tail -30
export Y2_TUNE_CACHE=/tmp/y2_tune_ab.json
timeout 300 python tools/train_steady.py 6 6 > /dev/null 2>&1
for rep in 1 2; do
  for pr in 1 0; do
    echo -n "prune=$pr: " >> $O/train_ab.log; Y2_PRUNE_OPERANDS=$pr timeout 300 python tools/train_steady.py 40 8 2>/dev/null | tail -1 >> $O/train_ab.log
  done
done
cat $O/train_ab.log
timeout 400 python tools/contention.py 2>/dev/null | tee $O/contention.txt | head -3
