#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_fullsize.py tests/test_gpu_round3.py -q --tb=short -rf -x 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/split_tests.log; tail -15 gpurun_out/split_tests.log
timeout 900 python bench.py --no-multiscale --no-conv3 --no-direct-leg --steps 30 --cpu-sample 0 > gpurun_out/bench_split.log 2> gpurun_out/bench_split.err; tail -3 gpurun_out/bench_split.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench_split.log') if l.startswith('{')][-1])
print(json.dumps(d['summary'], indent=0))
print(json.dumps(d['detect_kernel_table'].get('split_bf16x6'), indent=0)[:1500])
PY
Y2_SPLIT_BF16=1 timeout 600 python bench.py --no-multiscale --no-conv3 --no-direct-leg --no-detect --cpu-sample 0 > gpurun_out/bench_split_train.log 2> gpurun_out/bench_split_train.err; tail -3 gpurun_out/bench_split_train.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench_split_train.log') if l.startswith('{')][-1])
print('TRAIN with Y2_SPLIT_BF16=1:', d['train']['images_per_sec'], d['train']['ms_per_step'])
for r in d['train']['roofline']['top_kernels'][:14]: print(r['kernel'], r['launches_per_step'], r['ms_per_step'], r['frac'])
PY
