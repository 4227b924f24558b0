#!/bin/bash
# Round-3 final artefacts: full GPU test suite, smoke, default bench line, steady-state rocprofv3 profiles, contention rehearsal, glue count.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/pytest_gpu_full.log; tail -6 gpurun_out/pytest_gpu_full.log
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "=== bench"; timeout 1500 python bench.py > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log; tail -c 1500 gpurun_out/bench_stderr.log; grep -E '^\{' gpurun_out/bench_stdout.log | tail -1 > gpurun_out/bench_r03.json; tail -c 2600 gpurun_out/bench_r03.json; echo
echo "=== profiles"; bash tools/gpu_profile_r3.sh 2>&1 | tail -8
echo "=== contention"; timeout 600 python tools/contention.py 2>&1 | tee gpurun_out/contention.log | tail -3
echo "=== glue"; timeout 600 python tools/glue_count.py --sizes 320,416 > gpurun_out/glue.log 2>&1; grep -E "^S=" gpurun_out/glue.log
