#!/bin/bash
# pytest -m gpu (whole suite, no -x), log filtered of the ConfigChannels width warnings.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${PYTEST_TIMEOUT:-1800} python -m pytest tests -q -m gpu --tb=short ${PYTEST_ARGS} 2>&1 | grep -v "^WARNING:root" | tail -${PYTEST_TAIL:-150} | tee gpurun_out/pytest_gpu.log
