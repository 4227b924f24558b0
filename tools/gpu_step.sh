cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -u -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -rf --timeout=300 -k "detect_batch or postprocess" 2>&1 | grep --line-buffered -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -4
timeout 600 python bench.py --no-train --no-multiscale --no-conv3 --no-latency --no-resnet --no-direct-leg --no-split-leg --cpu-sample 0 --headline detect --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r.get('detect_to_host_images_per_sec'), r.get('detect_serial_images_per_sec'))"
