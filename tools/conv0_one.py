#!/usr/bin/env python
"""One first-layer launch per mode (for rocprofv3 --pmc runs): python tools/conv0_one.py [train|pool]"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else 'pool'
for _ in range(5):
    if mode == 'train':
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(w), None, None, _hip.ptr(z), None, _hip.ptr(stats), B, S, S, 3, C, C, 0, 1.0, st), 'c0')
    else:
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(sc), _hip.ptr(sh), None, _hip.ptr(yp), None, B, S, S, 3, C, 0, C, 0.1, st), 'c0')
torch.cuda.synchronize()
