#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Y2_BWD_STREAMS=1 timeout 600 python tools/train_table.py 2>&1 | grep -v "^WARNING" | tee gpurun_out/train_table.txt | tail -80
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q --tb=short -rf -k "608 or training" -s 2>&1 | grep -v "^WARNING\|WARNING  root" | tail -8
