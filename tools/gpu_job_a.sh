#!/bin/bash
# round-5 GPU call 1: new parity tests, the driver's bench command, batch-1 split-K probe
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/fa; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_b64.py "tests/test_gpu_plan.py::test_full_width_replays_equal_the_autograd_step" "tests/test_gpu_plan.py::test_a_parameter_frozen_after_the_first_steps_drops_the_plans" -x -q -s > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -30 $O/tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.log 2> $O/bench_stderr.log
echo "bench rc=$? line bytes: $(tail -1 $O/bench_stdout.log | wc -c)"
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
tail -1 $O/bench_stdout.log
tail -5 $O/bench_stderr.log
for s in 256 512 768 1024; do
  Y2_SPLIT_SLOTS=$s timeout 200 python tools/latency_b1.py 1 2 >> $O/latency.log 2>> $O/latency.err
done
cat $O/latency.log | cut -c1-400
