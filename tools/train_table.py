#!/usr/bin/env python
"""Per-layer, per-kernel time table of ONE Darknet-19 training step (B=64, 416x416) from the library's event hooks
(y2_prof_*, tags set by model/train_graph.py: 1+i = forward of block i, 101+i = backward of block i, 0 = loss / optimizer)."""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402
import bench_data  # noqa: E402
import train as y2train  # noqa: E402
import utils  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
y2train.GRAPH = False          # the event hooks bracket launches: the step's launch sequence issued eagerly (what the captured graph replays)
dev = torch.device('cuda:0')
inf, anchors = bench_data.build_model(20, dev, 'darknet')
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
data = {k: v.to(dev) for k, v in bench_data.labels(B, 416, 20, seed=2).items()}
data['tensor'] = bench_data.images(B, 416, seed=11).to(dev)
for _ in range(3):
    y2train.iterate(inf, opt, data, bench_data.HPARAM, 0.6, anchors)
torch.cuda.synchronize()
L = _hip.lib()
L.y2_prof_enable(1)
y2train.iterate(inf, opt, data, bench_data.HPARAM, 0.6, anchors)
torch.cuda.synchronize()
L.y2_prof_enable(0)
name = ctypes.create_string_buffer(96)
ms, fl = ctypes.c_float(), ctypes.c_double()
b1, b2, b3 = inf.dnn._blocks()
names = [n for n, _, _ in b1] + ['passthrough'] + [n for n, _, _ in b2] + [n for n, _, _ in b3]
rows = collections.OrderedDict()
for i in range(L.y2_prof_count()):
    L.y2_prof_get(i, name, 96, ctypes.byref(ms), ctypes.byref(fl))
    tag = L.y2_prof_get_tag(i)
    key = (tag, name.value.decode())
    e = rows.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1
    e[1] += ms.value
    e[2] += fl.value
tot = collections.defaultdict(float)
for (tag, k), (n, t, f) in rows.items():
    where = 'other' if tag == 0 else ('fwd ' + names[tag - 1] if tag < 100 else 'bwd ' + names[tag - 101])
    tot[where] += t
    print('%-16s %-30s x%-2d %8.3f ms %s' % (where, k, n, t, ('%6.1f TF/s' % (f / t / 1e9)) if f > 0 else ''))
print('---- per layer')
for k, v in tot.items():
    print('%-16s %8.3f ms' % (k, v))
print('total kernel time %.3f ms' % sum(tot.values()))
