#!/usr/bin/env python
"""First-layer kernel in its output modes + the box's pure write / copy rates (what bounds the training-mode conv0: 1.4 GB of un-pooled z at batch 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip
L = _hip.lib(); d = torch.device('cuda:0'); st = _hip.stream()
B, S, C = 64, 416, 32
x = torch.randn(B, 3, S, S, device=d); w = torch.randn(C, 3, 3, 3, device=d) * 0.1
sc = torch.rand(C, device=d) + 0.5; sh = torch.randn(C, device=d) * 0.1
z = torch.empty(B, S, S, C, device=d); yp = torch.empty(B, S // 2, S // 2, C, device=d)
stats = torch.zeros(32 * 2 * C, dtype=torch.float64, device=d)

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
modes = {'z + stats (training)': (None, None, z, None, stats), 'z only': (None, None, z, None, None), 'stats only': (None, None, None, None, stats),
         'pooled only (inference)': (sc, sh, None, yp, None), 'z + pooled': (sc, sh, z, yp, None)}
for name, (a, b, y, p, s) in modes.items():
    ms = t(lambda: _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(a), _hip.ptr(b), _hip.ptr(y), _hip.ptr(p), _hip.ptr(s), B, S, S, 3, C, C, C, 0.1, st), 'c0'))
    out = (z.numel() * 4 if y is not None else 0) + (yp.numel() * 4 if p is not None else 0)
    print('%-26s %.3f ms  (%.2f GB written -> %.2f TB/s)' % (name, ms, out / 1e9, out / ms / 1e9))
big = torch.empty(z.numel(), device=d)
print('fill  1.42 GB: %.3f ms' % t(lambda: big.zero_()))
print('copy  1.42 GB: %.3f ms (read + write)' % t(lambda: big.copy_(z.view(-1))))
