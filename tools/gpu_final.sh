#!/bin/bash
# Round artefacts.  Stages (all by default, or name them): tune tests smoke profiles bench driver contention
#   tune        tools/make_tune_table.py -> yolo2-pytorch_amd/tune/default_gfx950.json (copied to gpurun_out/ for committing)
#   tests       full GPU test suite            smoke   __graft_entry__.smoke()
#   profiles    rocprofv3 stats + PMC passes (tools/gpu_profile.sh) -> gpurun_out/prof/
#   bench       the default bench line -> gpurun_out/bench_r06.json (+ the long form gpurun_out/bench_r06_full.json)
#   driver      the driver's own command (`bench.py --gpus 1 --steps 20 --warmup 5`) -> gpurun_out/bench_r06_driver.json
#   contention  tools/contention.py -> gpurun_out/r06_contention.txt
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
STAGES="${*:-tune tests smoke profiles bench driver}"
for stage in $STAGES; do case $stage in
tune) echo "=== tune table"; timeout 1500 python tools/make_tune_table.py 2>&1 | grep -v amdgpu.ids | tail -4; cp yolo2-pytorch_amd/tune/default_gfx950.json gpurun_out/default_gfx950.json;;
tests) echo "=== pytest -m gpu"; timeout ${TESTS_TIMEOUT:-1500} python -u -m pytest tests -q -m gpu --tb=short -rf --durations=8 --timeout=400 ${PYTEST_ARGS} 2>&1 | grep --line-buffered -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" > gpurun_out/pytest_gpu_full.log; tail -22 gpurun_out/pytest_gpu_full.log;;
smoke) echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1;;
bench) echo "=== bench"; timeout 1500 python bench.py --tables gpurun_out/bench_r06_full.json > gpurun_out/bench_stdout.log 2> gpurun_out/bench_stderr.log; grep -v amdgpu.ids gpurun_out/bench_stderr.log | tail -c 1500
  tail -1 gpurun_out/bench_stdout.log > gpurun_out/bench_r06.json; echo "line bytes: $(wc -c < gpurun_out/bench_r06.json)"; cat gpurun_out/bench_r06.json;;
driver) echo "=== driver command"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --tables gpurun_out/bench_r06_driver_full.json 2>/dev/null | tail -1 > gpurun_out/bench_r06_driver.json; echo "line bytes: $(wc -c < gpurun_out/bench_r06_driver.json)"
  python -c "
import json; r = json.load(open('gpurun_out/bench_r06_driver.json'))['roofline']; print({k: r[k] for k in r if k.startswith(('multiscale', 'train_ms', 'train_images', 'detect_images', 'frac', 'traffic'))})
f = json.load(open('gpurun_out/bench_r06_driver_full.json'))['multiscale']; print([(p['size'], p['first_visit_ms'], p['first_visit_shapes_measured']) for p in f['per_size']], f.get('first_visit_measured_keys'), f.get('reserved_gib_before_after'))";;
profiles) echo "=== profiles"; bash tools/gpu_profile.sh 2>&1 | tail -40
  for f in detect_b32_traffic.json train_b64_traffic.json; do cp gpurun_out/prof/$f profiles/r06_$f; done;;      # a bench stage that follows reports this build's traffic
contention) echo "=== contention"; timeout 600 python tools/contention.py 2>/dev/null | tee gpurun_out/r06_contention.txt | head -3;;
esac; done
