#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_train.py -q --tb=short -rf 2>&1 | grep -v "^WARNING\|WARNING  root" | tail -12
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
for f in 1 0; do Y2_FUSE_CONV0=$f timeout 600 python bench.py --no-multiscale --no-conv3 --no-detect --cpu-sample 0 --train-steps 12 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['train']
print('Y2_FUSE_CONV0=$f train', t['images_per_sec'], 'img/s', t['ms_per_step'], 'ms; kernels:', ' '.join('%s=%.3f' % (r['kernel'], r['ms_per_step']) for r in t['roofline']['top_kernels'] if 'conv0' in r['kernel'] or 'bn_act_bwd' in r['kernel']))"; done
