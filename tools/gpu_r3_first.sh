#!/bin/bash
# Round-3 GPU call: parity tests, smoke, glue count, default bench line.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf ${PYTEST_ARGS:--x} 2>&1 | grep -v "^WARNING:root" > gpurun_out/pytest_gpu_full.log; tail -${PYTEST_TAIL:-40} gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu.log
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== glue"; timeout 600 python tools/glue_count.py > gpurun_out/glue.log 2>&1; grep -E "^S=|^ +[0-9]" gpurun_out/glue.log | head -60
echo "=== bench"; timeout 1200 python bench.py ${BENCH_ARGS} > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log; tail -c 2000 gpurun_out/bench_stderr.log; grep -E '^\{' gpurun_out/bench_stdout.log | tail -1 > gpurun_out/bench_r03.json; tail -c 2500 gpurun_out/bench_r03.json
