cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; export Y2_TUNE_STALE_OK=1
echo "=== wino6 mosaic tests"; timeout 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_plan.py -q -m gpu --tb=short -rf --timeout=300 -k "f43 or winograd_wgrad or wino or replay_safe" 2>&1 | grep --line-buffered -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -15
