cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; export Y2_TUNE_STALE_OK=1
echo "=== arena probe (reserve, all sizes)"; timeout 900 python tools/arena_probe.py reserve > gpurun_out/arena_reserve.json 2> gpurun_out/arena_reserve.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/arena_reserve.json'))
print({k:v for k,v in d.items() if k not in ('per_size',)})
for r in d['per_size']: print(r['size'], r['first_visit_ms'], r['ms_per_step'], r['reserved_gib'], r['allocated_gib'], r['snapshot'])
PY
grep -v amdgpu.ids gpurun_out/arena_reserve.err | tail -10
echo "=== arena tests"; timeout 600 python -u -m pytest tests/test_gpu_plan.py -q -m gpu --tb=short -rf --timeout=300 -k "arena or share_one_pool or recaptured or frozen or falls_back" 2>&1 | grep --line-buffered -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -15
