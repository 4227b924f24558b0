#!/bin/bash
# Round-2 first GPU call: parity tests, smoke, default bench line.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf ${PYTEST_ARGS:--x} 2>&1 | grep -v "^WARNING:root" > gpurun_out/pytest_gpu_full.log; tail -${PYTEST_TAIL:-40} gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu.log
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== bench"; timeout 900 python bench.py > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log; tail -c 3000 gpurun_out/bench_stderr.log; grep -E '^\{' gpurun_out/bench_stdout.log | tail -1 > gpurun_out/bench_r02.json; head -c 6000 gpurun_out/bench_r02.json
