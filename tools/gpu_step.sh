cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_final.sh tests smoke driver bench
