#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --multiscale --ms-sizes 320,416,608 --ms-maintain 4 2> gpurun_out/ms.err | tail -c 900; tail -3 gpurun_out/ms.err
echo; timeout 900 python bench.py --model resnet50 --size 608 --classes 80 --batch 16 --train-batch 16 --steps 10 --train-steps 4 --cpu-sample 0 2> gpurun_out/rn.err | tail -c 700; tail -3 gpurun_out/rn.err
echo; timeout 600 python bench.py --model tiny --steps 10 --train-steps 4 --cpu-sample 0 2> gpurun_out/tiny.err | tail -c 500; tail -3 gpurun_out/tiny.err
