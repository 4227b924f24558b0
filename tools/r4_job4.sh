#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== f34 layers B=4"; timeout 300 python tools/debug/f34_layers.py 4 2>&1 | grep -v amdgpu.ids
echo "=== latency leg: default / Winograd off"
for env in "Y2_X=1" "Y2_WINOGRAD=0"; do
env $env timeout 600 python bench.py --no-train --no-conv3 --no-multiscale --cpu-sample 0 --no-resnet --no-direct-leg --no-split-leg --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)
        for B in ('b1','b8'):
            e = r['latency'][B]; print('$env', B, e['ms_per_step'], e['launches_per_step'], e['kernel_ms_sum_eager'], e['winograd_layers'])
"
done
