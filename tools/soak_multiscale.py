#!/usr/bin/env python
"""Soak test of the captured training steps: COCO-80 Darknet-19, batch 64, the ten multi-scale sizes in random order with a random box count per batch (so that
plans for several padded label sizes coexist), a few hundred steps.  Prints reserved memory after the warm-up and at the end (the shared graph pool must not
grow without bound), the number of plans / captures, throughput, and that the loss stays finite."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch

import bench_data
import train as y2train
import utils

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda', 0)
sizes = [int(v) for v in os.environ['SOAK_SIZES'].split(',')] if os.environ.get('SOAK_SIZES') else [320 + 32 * i for i in range(10)]
inf, anchors = bench_data.build_model(80, dev, 'darknet')
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-4, momentum=0.9)
rng = random.Random(0)
data = {}
for S in sizes:
    for nmax in (6, 12, 30):
        d = {k: v.to(dev) for k, v in bench_data.labels(64, S, 80, nmax=nmax, seed=S + nmax).items()}
        d['tensor'] = bench_data.images(64, S, seed=S).to(dev)
        data[S, nmax] = d
keys = list(data)
mem = []
t0 = time.time()
S = sizes[0]
for i in range(steps):
    if i % 10 == 0:
        S = rng.choice(sizes)                      # utils/data.py:135-141: a new size every `maintain` batches
    r = y2train.iterate(inf, opt, data[S, rng.choice((6, 12, 30))], bench_data.HPARAM, bench_data.THRESHOLD, anchors)
    if os.environ.get('SOAK_CHECK'):
        rn = inf.__dict__.get('_y2_step_runner')
        bad = [k for k, q in inf.dnn.named_parameters() if not torch.isfinite(q).all()] + ['grad:' + k for k, q in inf.dnn.named_parameters() if q.grad is not None and not torch.isfinite(q.grad).all()]
        bad += ['buf:' + k for k, q in inf.dnn.named_buffers() if q.dtype.is_floating_point and not torch.isfinite(q).all()]
        lt = float(r['loss_total'].detach())
        print(i, S, getattr(rn, 'last', None), 'captures', None if rn is None else rn.captures, 'plans', None if rn is None else len(rn.plans), 'loss', round(lt, 5), 'nonfinite', bad[:6], flush=True)
        if bad or lt != lt:
            sys.exit(1)
    if os.environ.get('SOAK_TRACE') and (i < 60 or i % 10 == 0):
        lt = float(r['loss_total'].detach())
        print(i, S, round(lt, 5), [round(float(r['loss'][k].detach()), 5) for k in r['loss']], flush=True)
        if lt != lt:
            sys.exit(1)
    if i % 50 == 49:
        torch.cuda.synchronize()
        lt = float(r['loss_total'].detach())
        assert lt == lt and abs(lt) < 1e6, (i, lt)
        mem.append((i + 1, round(torch.cuda.memory_reserved() / 2 ** 30, 2), round(torch.cuda.memory_allocated() / 2 ** 30, 2), round(lt, 5), round(time.time() - t0, 1)))
runner = inf.__dict__['_y2_step_runner']
print(json.dumps({'steps': steps, 'plans': len(runner.plans), 'captures': runner.captures, 'broken': runner.broken, 'reserved_allocated_GiB_loss_seconds': mem}))
