#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest "tests/test_gpu_fullsize.py::test_full_width_training_step_with_the_gradient_algorithms_pinned" tests/test_gpu_train.py tests/test_gpu_plan.py tests/test_gpu_round3.py "tests/test_gpu_fullsize.py::test_608_coco80_full_width_training_step_against_the_oracle_and_its_fp32_floor" -q --tb=line -s 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | grep -i "passed\|failed\|error \|assert\|worst g\|pinned" | tail -30
echo "=== resnet leg"; timeout 900 python bench.py --no-detect --no-conv3 --no-multiscale --cpu-sample 0 --no-latency --train-steps 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); t = r['resnet50_608']['train']
        print({k: v for k, v in t.items() if k != 'roofline'})
        for k in t['roofline']['top_kernels'][:12]: print('   ', k)
"
