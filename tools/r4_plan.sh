#!/bin/bash
# Round 4, job: the StepPlan tests, then the default bench line (with the new latency / ResNet-50 / CPU legs).
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== plan tests"; timeout 1500 python -m pytest tests/test_gpu_plan.py -q --tb=short 2>&1 | grep -v "^WARNING\|WARNING  root" | tail -40
echo "=== bench"; timeout 1500 python bench.py > gpurun_out/r4_bench_stdout.log 2> gpurun_out/r4_bench_stderr.log; tail -c 1500 gpurun_out/r4_bench_stderr.log; grep -E '^\{' gpurun_out/r4_bench_stdout.log | tail -1 > gpurun_out/r4_bench.json; python - <<'PY'
import json
r = json.load(open('gpurun_out/r4_bench.json'))
print(json.dumps(r['summary'], indent=0))
print(json.dumps(r.get('cpu_baseline'), indent=0))
for B in ('b1', 'b8'):
    e = r['latency'][B]
    print(B, {k: v for k, v in e.items() if k not in ('plan', 'top_kernels')})
    print('  plan:', [(l['layer'], l['algo'], l['tile']) for l in e['plan']])
    for k in e['top_kernels']: print('   ', k)
print(r['resnet50_608']['detect'], {k: v for k, v in r['resnet50_608']['train'].items() if k != 'roofline'})
for k in r['resnet50_608']['train']['roofline']['top_kernels']: print('   ', k)
PY
