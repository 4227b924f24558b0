#!/bin/bash
# First-contact run: each kernel family in its own process (a GPU fault only kills that group), no -x.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx9" | head -6
nproc
for k in "conv_fwd" "conv0 or maxpool" "darknet" "decode or iou or nms or postprocess or detect_batch"; do
  echo "=== pytest -k '$k'"
  timeout 600 python -m pytest tests -q -m gpu --tb=short -k "$k" 2>&1 | grep -vE "^\s*$" | tail -40
done 2>&1 | tee gpurun_out/pytest_gpu.log
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== bench"; timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 8 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "=== layer bench"; timeout 600 python tools/layer_bench.py --reps 3 2>&1 | tail -100 | tee gpurun_out/layer_bench.log
