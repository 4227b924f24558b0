#!/bin/bash
# One gpurun call = parity tests + smoke + bench + per-layer sweep + rocprof kernel trace.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu --tb=short -x 2>&1 | tail -${PYTEST_TAIL:-60} | tee gpurun_out/pytest_gpu.log
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== bench"; timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
if [ -n "$LAYER_BENCH" ]; then echo "=== layer bench"; timeout 600 python tools/layer_bench.py $LAYER_BENCH 2>&1 | tail -100 | tee gpurun_out/layer_bench.log; fi
if [ -n "$ROCPROF" ]; then
  echo "=== rocprofv3 kernel trace"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-sample 0 > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
  tail -3 gpurun_out/rocprof.log
  find gpurun_out/prof -name '*stats*' | head; f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f"
fi
