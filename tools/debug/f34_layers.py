#!/usr/bin/env python
"""Per-layer check of the gradient algorithms at the full Darknet-19 shapes (416x416, batch 4): weight gradient by the direct kernel, the
2x2-tile and the 4x4-tile Winograd reductions, data gradient by the direct kernel and Winograd F(4x4,3x3) - each against the direct
kernel's result (which the single-layer tests hold to fp64).  Written to find which layer breaks when every eligible layer is pinned."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch

import _hip

L = _hip.lib()
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LAYERS = [(208, 32, 64), (104, 64, 128), (104, 64, 128), (52, 128, 256), (52, 128, 256), (26, 256, 512), (26, 256, 512), (13, 512, 1024), (13, 1024, 1024), (13, 1280, 1024)]


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().pow(2).mean().sqrt().clamp_min(1e-30)).item()


for HW, cin, cout in LAYERS:
    g = torch.Generator(device='cpu').manual_seed(HW + cin)
    x = torch.randn(B, HW, HW, cin, generator=g).to(d)
    dz = (torch.randn(B, HW, HW, cout, generator=g) * 1e-3).to(d)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(d)
    ref = torch.zeros(cout * 9 * cin, device=d)
    _hip.check(L.y2_conv_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(ref), B, HW, HW, cin, cin, cout, cout, 3, _hip.stream()), 'direct')
    refn = torch.empty(cout, cin, 3, 3, device=d)
    _hip.check(L.y2_unpack_weight_grad(_hip.ptr(ref), _hip.ptr(refn), cout, cin, 3, _hip.stream()), 'unpack')
    need = L.y2_wino_wgrad_workspace_bytes(B, HW, HW, cin, cout)
    ws = torch.empty(need // 4 + 4, device=d)
    out = []
    for flags in (0, 1, 2, 3):
        t = torch.full((cout * cin * 9,), 3.0, device=d)
        rc = L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(t), B, HW, HW, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, flags, _hip.stream())
        if rc != 0:
            out.append('flags%d rc=%d' % (flags, rc))
            continue
        got = t.view(cout, cin, 3, 3) if flags & 1 else None
        if got is None:
            got = torch.empty(cout, cin, 3, 3, device=d)
            _hip.check(L.y2_unpack_weight_grad(_hip.ptr(t), _hip.ptr(got), cout, cin, 3, _hip.stream()), 'unpack')
        out.append('flags%d %.1e' % (flags, rel(got, refn)))
    # data gradient: direct vs F(4x4,3x3)
    wd = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wd), cout, cin, 3, 1, _hip.stream()), 'pack1')
    res = {}
    for algo in (0, 6):
        dx = torch.empty(B, HW, HW, cin, device=d)
        p = _hip.ConvParams()
        wop = wd if algo == 0 else _hip.wino6_weight(wd, cin, cout)
        p.x, p.w, p.y, p.algo = dz.data_ptr(), wop.data_ptr(), dx.data_ptr(), algo
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope = B, HW, HW, cout, cout, cin, 3, cin, 1.0
        p.tile = 5 if algo == 6 else 0
        if _hip.conv_workspace(p, d) < 0:
            res[algo] = None
            continue
        rc = L.y2_conv_fwd(ctypes.byref(p), _hip.stream())
        res[algo] = dx if rc == 0 else None
    dg = 'f43 %s' % ('n/a' if res.get(6) is None else '%.1e' % rel(res[6], res[0]))
    torch.cuda.synchronize()
    print('%3dx%-3d %4d->%-4d  wgrad vs direct: %s | dgrad %s' % (HW, HW, cin, cout, '  '.join(out), dg), flush=True)
