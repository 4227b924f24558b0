#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_fullsize.py -q --tb=short -rf -x -s 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/split_tests.log; grep -oE "forced plan.*|[0-9]+ (passed|failed).*|FAILED.*|Error.*" gpurun_out/split_tests.log | sort -u | head -30
timeout 900 python bench.py --no-multiscale --no-conv3 --no-direct-leg --no-train --steps 40 --cpu-sample 0 > gpurun_out/bench_split.log 2> gpurun_out/bench_split.err; tail -2 gpurun_out/bench_split.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench_split.log') if l.startswith('{')][-1])
print({k: v for k, v in d['summary'].items() if 'split' in k or 'detect' in k})
PY
Y2_SPLIT_F16=1 timeout 600 python tools/train_steady.py 16 2>/dev/null | tail -1 | sed 's/^/Y2_SPLIT_F16=1 train /'
