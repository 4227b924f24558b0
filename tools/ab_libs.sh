#!/bin/bash
# A/B of two builds of libyolo2_hip.so (same ABI): per-layer direct-conv sweep and the bench line with each.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
A=$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip.so
B=${1:-$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip_spread.so}
for lib in $A $B; do
  echo "=== $(basename $lib): layer_bench (direct kernels, B=32)"
  Y2_LIB=$lib python tools/layer_bench.py --tiles ${TILES:-1,2,5} --reps 5 2>&1 | grep -v "^$" | tail -60 > gpurun_out/layer_$(basename $lib .so).txt
done
paste -d'|' gpurun_out/layer_$(basename $A .so).txt gpurun_out/layer_$(basename $B .so).txt | cut -c1-200 | tail -60
for lib in $A $B; do
  echo "=== $(basename $lib): bench"
  Y2_LIB=$lib python bench.py --steps 50 --no-conv3 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('detect', d['detect']['images_per_sec'], 'img/s', d['detect']['ms_per_step'], 'ms;  direct-only', d['roofline']['direct_only'].get('images_per_sec'), ' train', d['train']['images_per_sec'], 'img/s', d['train']['ms_per_step'], 'ms')
for r in d['roofline']['top_kernels'][:6]: print('   ', r['kernel'], r['ms_per_step'], r['executed_tflops'], r['frac'])
for r in d['train']['roofline']['top_kernels'][:8]: print('   T', r['kernel'], r['ms_per_step'], r['executed_tflops'], r['frac'])
"
done
