#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== tests"; timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_train.py "tests/test_gpu_fullsize.py::test_full_width_training_step_with_the_gradient_algorithms_pinned" tests/test_gpu_dp.py::test_dp_world2_darknet_region_loss_equals_concatenated_batch -q --tb=short -s 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | grep -i "passed\|failed\|error\|assert\|worst\|pinned" | tail -40
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "=== resnet leg"; timeout 900 python bench.py --no-detect --no-conv3 --no-multiscale --cpu-sample 0 --no-latency --train-steps 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); t = r['resnet50_608']['train']
        print({k: v for k, v in t.items() if k != 'roofline'})
        for k in t['roofline']['top_kernels'][:8]: print('   ', k)
"
