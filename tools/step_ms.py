#!/usr/bin/env python
"""ms per captured step of the two timed workloads with the library and table in force: the batch-64 training step and the batch-32 detect step.
A/B runs: Y2_LIB=<other build> Y2_TUNE_STALE_OK=1 python tools/step_ms.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip, bench_data, detect, train as y2train, utils

dev = torch.device('cuda', 0)
inf, anchors = bench_data.build_model(20, dev, 'darknet')


def timed(fn, n, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


inf.eval()
x = bench_data.images(32, 416, seed=1).to(dev)
run = detect.GraphedDetector(inf.dnn, anchors, x, static_input=True, fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
for _ in range(5):
    run.run()
torch.cuda.synchronize()
d = timed(run.run, 50)
inf.train()
opt = utils.optim.SGD(inf.parameters(), 0.0, momentum=0.9)
data = {k: v.to(dev) for k, v in bench_data.labels(64, 416, 20, seed=2).items()}
data['tensor'] = bench_data.images(64, 416, seed=11).to(dev)
step = lambda: y2train.iterate(inf, opt, data, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
for _ in range(8):
    step()
torch.cuda.synchronize()
t = timed(step, 30)
print('%s: detect b32 %.3f ms (serial replays)   train b64 %.3f ms' % (os.path.basename(os.path.dirname(_hip.LIB_PATH)), d, t))
