#!/bin/bash
# A/B of the BatchNorm / activation kernels' grid (items per thread, workgroups per CU): per-kernel sums of one training step (single stream)
cd "$(dirname "$0")/.." || exit 1
export Y2_TUNE_DEFAULTS=0 Y2_TUNE_CACHE=/tmp/tc.json Y2_BWD_STREAMS=1
python tools/train_table.py 64 > /dev/null 2>&1
for cfg in "8 8" "8 4" "8 16" "4 16" "16 8" "16 4" "32 2"; do
  set -- $cfg
  Y2_ACT_ITEMS=$1 Y2_ACT_WG_PER_CU=$2 python tools/train_table.py 64 2>/dev/null > /tmp/t.txt
  echo "items=$1 wg_per_cu=$2: bn_act_bwd $(grep 'bn_act_bwd_kernel' /tmp/t.txt | awk '{s+=$(NF-1)} END {print s}') ms, bn_act_fwd $(grep 'bn_act_fwd_kernel' /tmp/t.txt | awk '{s+=$(NF-1)} END {print s}') ms, block_reduce $(grep 'bn_bwd_block_reduce' /tmp/t.txt | awk '{s+=$(NF-1)} END {print s}') ms, $(tail -1 /tmp/t.txt)"
done
