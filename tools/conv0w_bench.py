#!/usr/bin/env python
"""Time y2_conv0_wgrad (first-layer weight gradient, B=64, 416x416, 3 -> 32) and check it against torch autograd on a small case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip
L, st = _hip.lib(), _hip.stream()
dev = torch.device('cuda:0')
B, H, W, cin, cout = 64, 416, 416, 3, 32
x = torch.randn(B, cin, H, W, device=dev)
dz = torch.randn(B, H, W, cout, device=dev)
dw = torch.zeros(cout, cin, 3, 3, device=dev)
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _hip.check(L.y2_conv0_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dw), B, H, W, cin, cout, cout, st), 'conv0_wgrad')
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print('conv0_wgrad B=64: %.4f ms' % best)
xs = torch.randn(2, 3, 20, 36, device=dev, dtype=torch.float64, requires_grad=True)
ws = torch.randn(32, 3, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
g = torch.randn(2, 32, 20, 36, device=dev, dtype=torch.float64)
torch.nn.functional.conv2d(xs, ws, padding=1).backward(g)
dws = torch.zeros(32, 3, 3, 3, device=dev)
_hip.check(L.y2_conv0_wgrad(_hip.ptr(xs.detach().float().contiguous()), _hip.ptr(g.permute(0, 2, 3, 1).float().contiguous()), _hip.ptr(dws), 2, 20, 36, 3, 32, 32, st), 'conv0_wgrad')
torch.cuda.synchronize()
print('max rel err %.2e' % ((dws.double() - ws.grad).abs().max() / ws.grad.abs().max()).item())
