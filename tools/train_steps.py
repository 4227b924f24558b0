#!/usr/bin/env python
"""Per-step wall time of the Darknet-19 training step (B=64, 416x416) over N steps: drift / allocator / clock effects."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import bench_data, train as y2train, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
inf, anchors = bench_data.build_model(20, dev, 'darknet')
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
data = {k: v.to(dev) for k, v in bench_data.labels(64, 416, 20, seed=2).items()}
data['tensor'] = bench_data.images(64, 416, seed=11).to(dev)
ts = []
for i in range(n + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y2train.iterate(inf, opt, data, bench_data.HPARAM, 0.6, anchors)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('ms per step:', ' '.join('%.1f' % t for t in ts))
print('allocated GB %.1f reserved GB %.1f' % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
