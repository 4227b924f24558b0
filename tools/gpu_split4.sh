#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py -q --tb=short -rf -s 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/split_tests.log; grep -oE "gemm_split_f16 [0-9x ]+: .*|conv [0-9]+x[0-9]+ Cin.*|[0-9]+ (passed|failed).*|FAILED.*|Error.*" gpurun_out/split_tests.log | sort -u | head -40
timeout 600 python tools/split_bench.py 2>&1 | tee gpurun_out/split_bench.log | tail -8
