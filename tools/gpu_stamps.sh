#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
STAMPS_F16=1 Y2_LIB=$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip_stamps.so timeout 300 python tools/split_stamps.py 2>&1 | tee gpurun_out/split_stamps_f16.log | grep -E "BK=|K loop|mean"
timeout 300 python -m pytest tests/test_gpu_split.py -q -k overflow 2>&1 | tail -2
