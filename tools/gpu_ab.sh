#!/bin/bash
# A/B of one environment switch on the training leg: tools/gpu_ab.sh VAR
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_round3.py -q --tb=short -rf -x 2>&1 | grep -v "^WARNING\|WARNING  root" | tail -4
for v in 1 0 1 0; do env $1=$v timeout 600 python tools/train_steady.py 24 2>/dev/null | tail -1 | sed "s/^/$1=$v /"; done
