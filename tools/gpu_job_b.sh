#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export Y2_TUNE_DEFAULTS=0
for rep in 1 2; do for sk in 1 0; do
  echo -n "splitk=$sk: "; Y2_SPLITK=$sk Y2_TUNE_CACHE=/tmp/tc_$sk.json timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --no-train --no-direct-leg --no-conv3 --no-split-leg --no-multiscale --no-latency --no-resnet --tables /tmp/t.json 2>/dev/null | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.read())['roofline']; print(r['detect_images_per_sec'], r['detect_serial_images_per_sec'], r['conv_chain_ms_per_step'])"
done; done
