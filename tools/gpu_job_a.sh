#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/fj; mkdir -p $O
for rep in 1 2 3; do for pr in 1 0; do
  echo -n "prio=$pr: " >> $O/ab.log; Y2_CAPTURE_PRIO=$pr timeout 300 python tools/train_steady.py 40 8 2>/dev/null | tail -1 >> $O/ab.log
done; done
cat $O/ab.log | cut -c1-8,130-
