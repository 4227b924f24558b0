#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/fg; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_plan.py -x -q -s -k "capture_falls or frozen or darknet_step_plan or dp_world2" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -v "^WARNING\|amdgpu.ids" $O/tests.log | grep -i "passed\|failed\|error\|rc=" | tail -30
