#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -q --tb=short -rf -k "slots or first_layer" 2>&1 | grep -v "^WARNING\|WARNING  root" | tail -5
for st in 1 2 3; do timeout 600 python bench.py --no-multiscale --no-conv3 --no-train --no-direct-leg --cpu-sample 0 --steps 60 --streams $st --rotate 6 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams=$st detect', d['detect']['images_per_sec'], d['detect']['ms_per_step'], '| split', d['summary'].get('split_bf16x6_detect_images_per_sec'), d['detect']['launch'])"; done
