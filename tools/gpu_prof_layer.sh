#!/bin/bash
# PMC counters for one conv layer (separate passes; --pmc never combined with trace domains other than kernel-trace)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
LAYER=${LAYER:-l1.8}; TILE=${TILE:-1}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|OccupancyPercent)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/counters.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/tools/layer_bench.py --only $LAYER --tiles $TILE --reps 2 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'conv_fwd' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, 'per-dispatch mean', sum(v)/len(v), 'n', len(v))
PY
done 2>&1 | tee $R/gpurun_out/pmc_summary.txt
