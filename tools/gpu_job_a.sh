#!/bin/bash
# same-box A/B of two algorithm tables (tools/probes/tune_old.json / tune_new.json as Y2_TUNE_CACHE, default table off)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/fi; mkdir -p $O
for rep in 1 2 3; do for t in old new; do
  cp tools/probes/tune_$t.json /tmp/tc_$t.json
  echo -n "$t: " >> $O/ab.log; Y2_TUNE_DEFAULTS=0 Y2_TUNE_CACHE=/tmp/tc_$t.json timeout 300 python tools/train_steady.py 40 8 2>/dev/null | tail -1 >> $O/ab.log
done; done
cat $O/ab.log | cut -c1-6,130-
