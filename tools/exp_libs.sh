#!/bin/bash
# kernel-only wf_bench under several experiment builds of the library (csrc/libyolo2_exp*.so)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for lib in yolo2-pytorch_amd/csrc/libyolo2_hip.so yolo2-pytorch_amd/csrc/libyolo2_exp*.so; do
  echo "=== $(basename $lib)"
  Y2_LIB=$PWD/$lib timeout 120 python tools/wf_bench.py --batch ${BATCH:-32} --variants=${VARIANTS:-3,100} --kernel-only --shapes ${SHAPES:-104x64x128,52x128x256,26x512x512,104x128x64} 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/exp_libs.txt
