#!/usr/bin/env python
"""Is a library call replay-safe?  Capture y2_wino_wgrad_ex (2x2 / 4x4 tiles) and y2_conv_wgrad into a hipGraph, replay three times on changing inputs and
compare every replay with an eager call on the same inputs."""
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
for (B, HW, cin, cout) in ((64, 52, 128, 256), (64, 26, 256, 512), (64, 104, 64, 128), (64, 13, 512, 1024)):
    x = torch.randn(B, HW, HW, cin, device=d)
    dz = torch.randn(B, HW, HW, cout, device=d) * 1e-3
    need = L.y2_wino_wgrad_workspace_bytes(B, HW, HW, cin, cout)
    ws = torch.empty(need // 4 + 4, device=d)
    for flags in (1, 3):
        out = torch.empty(cout, cin, 3, 3, device=d)
        ref = torch.empty(cout, cin, 3, 3, device=d)

        def call(dst):
            _hip.check(L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dst), B, HW, HW, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, flags, _hip.stream()), 'wgrad')
        call(ref)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            g.capture_begin()
            call(out)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        errs = []
        for rep in range(3):
            x.normal_()
            dz.normal_().mul_(1e-3)
            ws.fill_(float('nan') if rep == 1 else 1e30)          # whatever the scratch holds between two uses
            g.replay()
            torch.cuda.synchronize()
            got = out.clone()
            ws.zero_()
            call(ref)
            torch.cuda.synchronize()
            errs.append(((got - ref).abs().max() / ref.abs().max()).item())
        print('%dx%d %d->%d flags %d: replay vs eager rel err %s' % (HW, HW, cin, cout, flags, ['%.1e' % e for e in errs]), flush=True)
