#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf ${PYTEST_ARGS} 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/pytest_gpu_full.log; tail -12 gpurun_out/pytest_gpu_full.log
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "=== contention"; timeout 600 python tools/contention.py 2>&1 | tee gpurun_out/contention.log | tail -4
echo "=== glue"; timeout 600 python tools/glue_count.py --sizes 320,416 > gpurun_out/glue.log 2>&1; grep -E "^S=|^ +[0-9]" gpurun_out/glue.log | head -30
