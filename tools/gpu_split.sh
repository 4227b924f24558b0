#!/bin/bash
# split-bf16 mode: unit / conv / full-size parity tests, then the layer-level A/B.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_round3.py "tests/test_gpu_fullsize.py::test_full_batch_under_each_forced_algorithm" tests/test_gpu_kernels.py::test_resnet_matches_reference_fixture -q --tb=short -rf -s ${PYTEST_ARGS} 2>&1 | grep -v "^WARNING\|WARNING  root" > gpurun_out/split_tests.log; grep -E "gemm_split|^conv |forced plan|passed|failed|FAILED|Error|assert" gpurun_out/split_tests.log | head -80
timeout 600 python tools/split_bench.py 2>&1 | tee gpurun_out/split_bench.log | tail -12
