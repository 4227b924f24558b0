#!/usr/bin/env python
"""Steady-state training steps of a model.resnet plugin (BASELINE configs[4] per GPU: resnet50, 608x608, COCO-80, batch 32): the ResNet twin of tools/train_steady.py.

    python tools/resnet_steady.py [steps [warm-up steps [arch [size [batch]]]]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import bench_data, train as y2train, utils

arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
steps, warm, arch, S, B = arg(1, 10), arg(2, 6), arg(3, 'resnet50'), arg(4, 608), arg(5, 32)
dev = torch.device('cuda:0')
inf, anchors = bench_data.build_model(80, dev, arch)
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
d = {k: v.to(dev) for k, v in bench_data.labels(B, S, 80, seed=72).items()}
d['tensor'] = bench_data.images(B, S, seed=71).to(dev)
for i in range(warm):
    y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    r = y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({'arch': arch, 'size': S, 'batch': B, 'steps_timed': steps, 'ms_per_step': round(dt / steps * 1e3, 3), 'images_per_sec': round(B * steps / dt, 1), 'loss_total': float(r['loss_total'])}))
