#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest "tests/test_gpu_fullsize.py::test_full_width_training_step_with_the_gradient_algorithms_pinned" tests/test_gpu_train.py::test_darknet_training_step_matches_oracle_autograd tests/test_gpu_dp.py::test_dp_world2_darknet_region_loss_equals_concatenated_batch -q --tb=line -s 2>&1 | grep -v "^WARNING\|WARNING  root\|Gloo\|amdgpu.ids\|socket.cpp" | grep -i "passed\|failed\|error \|assert\|worst\|pinned\|ratio" | tail -60
