#!/usr/bin/env python
"""Weight-gradient candidates of one layer shape timed the way _hip.conv_wgrad times them (back-to-back launches) and with the caches evicted in between."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip
L = _hip.lib()
d = torch.device('cuda:0')
B, H, W, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 208, 208, 32, 64))]
x = torch.randn(B, H, W, cin, device=d)
dz = torch.randn(B, H, W, cout, device=d) * 1e-3
dwp = torch.zeros(cout * cin * 9, device=d)
need = L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout)
ws = torch.empty(need // 4 + 4, device=d)
evict = torch.empty(768 << 20, dtype=torch.uint8, device=d)
st = _hip.stream()
fns = {'direct': lambda: _hip.check(L.y2_conv_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, 3, st), 'd'),
       'wino2x2': lambda: _hip.check(L.y2_wino_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, st), 'w'),
       'wino4x4': lambda: _hip.check(L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, 2, st), 'w6')}
for name, fn in fns.items():
    fn(); torch.cuda.synchronize()
    hot = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); e1.synchronize()
        hot.append(e0.elapsed_time(e1) / 2)
    cold = []
    for _ in range(4):
        evict.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        cold.append(e0.elapsed_time(e1))
    print('%-8s back-to-back %.3f ms   evicted %.3f ms' % (name, min(hot), min(cold)))
