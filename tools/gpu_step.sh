cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== A/B switches off: plan / train tests"; Y2_TRAIN_ARENA=0 Y2_WINO6_TALL=0 Y2_FUSE_WINO6=0 timeout 1200 python -u -m pytest tests/test_gpu_plan.py tests/test_gpu_train.py -q -m gpu --tb=short -rf --timeout=300 -x --deselect tests/test_gpu_plan.py::test_plans_of_all_sizes_live_in_one_activation_arena_sized_for_the_largest 2>&1 | grep --line-buffered -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -6
echo "=== autograd path (no plans) b64-size step"; Y2_TRAIN_PLAN=0 timeout 300 python tools/train_steady.py 20 4 2>&1 | grep -v amdgpu.ids | tail -1
echo "=== linear graph"; Y2_GRAPH_FORK=0 timeout 300 python tools/train_steady.py 40 8 2>&1 | grep -v amdgpu.ids | tail -1
echo "=== single stream"; Y2_BWD_STREAMS=1 timeout 300 python tools/train_steady.py 40 8 2>&1 | grep -v amdgpu.ids | tail -1
