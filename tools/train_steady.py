#!/usr/bin/env python
"""Steady-state training steps and nothing else (the command the training rocprofv3 profiles are taken from): Darknet-19 VOC-20,
batch 64, 416x416, utils.optim.SGD; with Y2_TUNE_CACHE pre-populated no launch of the process is an algorithm-selection launch.

    Y2_TUNE_CACHE=/tmp/y2_tune_train.json python tools/train_steady.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import bench_data, train as y2train, utils

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # 3: what the rocprofv3 profiles use (Y2_TRAIN_GRAPH=0); 6 puts the capture of the step in front of the timed region
dev = torch.device('cuda:0')
inf, anchors = bench_data.build_model(20, dev, 'darknet')
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
data = []
for i in range(2):
    d = {k: v.to(dev) for k, v in bench_data.labels(64, 416, 20, seed=2 + i).items()}
    d['tensor'] = bench_data.images(64, 416, seed=11 + i).to(dev)
    data.append(d)
for i in range(warm):
    y2train.iterate(inf, opt, data[i % 2], bench_data.HPARAM, bench_data.THRESHOLD, anchors)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    y2train.iterate(inf, opt, data[i % 2], bench_data.HPARAM, bench_data.THRESHOLD, anchors)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
runner = inf.__dict__.get('_y2_step_runner')
plan = next(iter(runner.plans.values())) if (runner is not None and runner.plans) else None
info = {'captures': getattr(runner, 'captures', None), 'graph_ops': None if plan is None or plan.ops is None else len(plan.ops),
        'operand_forms_prepared': None if plan is None or plan.only is None else len(plan.only), 'capture_error': None if plan is None or plan.capture_error is None else str(plan.capture_error)[:200]}
print(json.dumps({'plan': info, 'steps_total': steps + warm, 'steps_timed': steps, 'ms_per_step': round(dt / steps * 1e3, 3), 'images_per_sec': round(64 * steps / dt, 1)}))
