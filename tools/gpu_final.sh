#!/bin/bash
# Round-4 artefacts.  Stages (all by default, or name them): tune tests smoke bench profiles pressure
#   tune      tools/make_tune_table.py -> yolo2-pytorch_amd/tune/default_gfx950.json (copied to gpurun_out/ for committing)
#   tests     full GPU test suite            smoke   __graft_entry__.smoke()
#   bench     the default bench line -> gpurun_out/bench_r04.json
#   profiles  rocprofv3 stats + PMC passes (tools/gpu_profile_r4.sh) -> gpurun_out/prof4/
#   pressure  tools/host_pressure.py -> gpurun_out/r4_host_pressure.jsonl
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
STAGES="${*:-tune tests smoke profiles bench}"
for stage in $STAGES; do case $stage in
tune) echo "=== tune table"; timeout 1200 python tools/make_tune_table.py 2>&1 | grep -v amdgpu.ids | tail -4; cp yolo2-pytorch_amd/tune/default_gfx950.json gpurun_out/default_gfx950.json;;
tests) echo "=== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu --tb=short -rf 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" > gpurun_out/pytest_gpu_full.log; tail -12 gpurun_out/pytest_gpu_full.log;;
smoke) echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1;;
bench) echo "=== bench"; timeout 1500 python bench.py > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log; grep -v amdgpu.ids gpurun_out/bench_stderr.log | tail -c 1500; grep -E '^\{' gpurun_out/bench_stdout.log | tail -1 > gpurun_out/bench_r04.json; python -c "
import json; r = json.load(open('gpurun_out/bench_r04.json')); print(json.dumps(r['summary'])); print(json.dumps({k: v for k, v in r['cpu_baseline'].items() if 'sample' not in k}))";;
profiles) echo "=== profiles"; bash tools/gpu_profile_r4.sh 2>&1 | tail -8
  for f in detect_b32_traffic.json train_b64_traffic.json; do cp gpurun_out/prof4/$f profiles/r04_$f; done;;      # a bench stage that follows reports this build's traffic
pressure) echo "=== host pressure"; timeout 1500 python tools/host_pressure.py 2>/dev/null | tee gpurun_out/r4_host_pressure.jsonl | tail -30;;
esac; done
