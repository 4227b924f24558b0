#!/usr/bin/env python
"""Contention rehearsal for the data-parallel training step on ONE GPU: RCCL's all-reduce kernels hold CUs while
the backward pass runs, and the persistent fused Winograd kernels need a whole CU per workgroup.  A background stream holds k workgroup slots
with a spin kernel (tools/probes/spin.hip: 256 threads + 32 KB LDS each, like a collective's channels) for the duration of every step; the
Darknet-19 training step (B=64, 416x416) is timed for k in {0, 8, 16, 32} with the fused kernels' tiles CLAIMED from per-XCD counters
(the shipped scheduler) and with a STATIC stride (Y2_WF_STATIC=1).

    bash tools/probes/build_spin.sh && python tools/contention.py"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import bench_data, train as y2train, utils

spin = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libspin.so'))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
dev = torch.device('cuda:0')
B, S = 64, 416
inf, anchors = bench_data.build_model(20, dev, 'darknet')
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
data = {k: v.to(dev) for k, v in bench_data.labels(B, S, 20, seed=2).items()}
data['tensor'] = bench_data.images(B, S, seed=11).to(dev)
step = lambda: y2train.iterate(inf, opt, data, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
for _ in range(6):
    step()
torch.cuda.synchronize()
side = torch.cuda.Stream()


def timed(k, reps=6, hold_ms=60.0):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if k:
            rc = spin.spin_launch(k, 256, hold_ms, ctypes.c_void_p(side.cuda_stream))
            assert rc == 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    ts.sort()
    return ts[len(ts) // 2]


out = {}
for mode in ('claimed', 'static'):
    os.environ['Y2_WF_STATIC'] = '1' if mode == 'static' else '0'
    for _ in range(2):
        step()
    out[mode] = {k: round(timed(k), 2) for k in (0, 8, 16, 32)}
    print('%-8s tiles: ms per training step with k CU slots held by a background kernel: %s' % (mode, '  '.join('k=%d: %.2f' % kv for kv in out[mode].items())))
print(json.dumps({'workload': 'Darknet-19 VOC-20 train step, batch 64, 416x416, single GPU, background spin kernel of k workgroups (256 threads, 32 KB LDS) for 60 ms per step', 'ms_per_step': out}))
