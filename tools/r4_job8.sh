#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_round3.py tests/test_gpu_plan.py -q --tb=short -x 2>&1 | grep -i "passed\|failed\|error\|assert" | tail -8
timeout 900 python bench.py --no-detect --no-conv3 --no-multiscale --cpu-sample 0 --no-latency --train-steps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); t = r['train']
        print({k: t.get(k) for k in ('images_per_sec', 'ms_per_step', 'host_issue_ms_per_step')})
        print([(k['kernel'], k['ms_per_step']) for k in t['roofline']['top_kernels'] if 'bn_' in k['kernel']])
        t = r['resnet50_608']['train']
        print({k: v for k, v in t.items() if k != 'roofline'})
        print([(k['kernel'], k['ms_per_step']) for k in t['roofline']['top_kernels'] if 'bn_' in k['kernel']])
"
