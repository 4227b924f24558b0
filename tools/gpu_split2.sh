#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/debug/unaligned_debug.py 2>&1 | grep -v "^WARNING" | tee gpurun_out/unaligned_debug.log
timeout 600 python -m pytest tests/test_gpu_split.py -q --tb=short -rf -s 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/split_tests.log; grep -oE "[0-9]+ (passed|failed).*|FAILED.*" gpurun_out/split_tests.log | head -40
Y2_LIB=$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip_stamps.so timeout 300 python tools/split_stamps.py 2>&1 | tee gpurun_out/split_stamps.log
timeout 600 python tools/split_bench.py 2>&1 | tee gpurun_out/split_bench.log | tail -8
