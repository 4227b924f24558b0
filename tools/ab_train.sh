#!/bin/bash
# A/B of two library builds on the training leg.  Usage: tools/ab_train.sh [other.so]
cd "$(dirname "$0")/.." || exit 1
A=$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip.so
B=${1:-$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip_b.so}
for lib in $A $B; do
  echo "=== $(basename $lib): train leg"
  Y2_LIB=$lib python bench.py --no-detect --no-conv3 --cpu-sample 0 --train-steps 12 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('train', d['train']['images_per_sec'], 'img/s', d['train']['ms_per_step'], 'ms')
for r in d['train']['roofline']['top_kernels'][:12]: print('   T', r['kernel'], r['launches_per_step'], r['ms_per_step'], r['executed_tflops'], r['frac'])
"
done
