cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_final.sh tune
echo "=== arena probe"; timeout 600 python tools/arena_probe.py reserve > gpurun_out/arena_reserve.json 2> gpurun_out/arena_reserve.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/arena_reserve.json'))
print({k:v for k,v in d.items() if k not in ('per_size',)})
for r in d['per_size']: print(r['size'], r['first_visit_ms'], r['ms_per_step'], r['reserved_gib'], r['allocated_gib'])
PY
grep -v amdgpu.ids gpurun_out/arena_reserve.err | tail -5
