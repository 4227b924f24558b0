#!/bin/bash
# Round 4, job 2: the StepPlan tests, the host-pressure rehearsal, the two-rank DP bench on one GPU.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== plan tests"; timeout 1500 python -m pytest tests/test_gpu_plan.py -q --tb=short 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -30
echo "=== host pressure"; nproc; timeout 1500 python tools/host_pressure.py 2> gpurun_out/r4_host_pressure.err | tee gpurun_out/r4_host_pressure.jsonl; grep -v amdgpu.ids gpurun_out/r4_host_pressure.err | tail -5
echo "=== dp tests"; timeout 1500 python -m pytest tests/test_gpu_dp.py -q --tb=short -x 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | tail -15
