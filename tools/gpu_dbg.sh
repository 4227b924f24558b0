#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/debug/taps.py 2>&1 | grep -v "^WARNING" | tee gpurun_out/taps.log | tail -30
