#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py::test_conv_fwd_matches_fp64_reference -q --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 600 python tools/persist_bench.py 2>&1 | grep -v amdgpu.ids
