cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_final.sh tests smoke
PROFILE_ROUND=r06 bash tools/gpu_profile.sh 2>&1 | tail -6
Y2_BWD_STREAMS=1 timeout 300 python tools/train_table.py 64 > gpurun_out/r06_train_b64_layer_table.txt 2>/dev/null; tail -1 gpurun_out/r06_train_b64_layer_table.txt
for f in detect_b32_traffic.json train_b64_traffic.json; do cp gpurun_out/prof/$f profiles/r06_$f; done
bash tools/gpu_final.sh driver bench contention
