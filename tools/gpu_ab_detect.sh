#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 600 python bench.py --no-multiscale --no-conv3 --no-train --no-direct-leg --no-split-leg --cpu-sample 0 --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', 'pipelined', d['detect']['images_per_sec'], 'serial', d['detect']['serial_images_per_sec'])"; }
run A=0
run Y2_WINO_IMPLICIT=0
run Y2_FORCE_ALGO=fused
run Y2_FORCE_ALGO=winograd
run Y2_FORCE_ALGO=implicit
run A=1
