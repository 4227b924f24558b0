cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 600 python -u -m pytest tests/test_gpu_dp.py -q -m gpu --tb=line -rf --timeout=300 -k "equals_concatenated" 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -2; done
