#!/usr/bin/env python
"""y2_conv0_fwd alone (first layer 3 -> 32, 416x416): ms per launch for the inference form (pooled output) and the training form
(raw output + statistics), from the library's per-kernel event hooks."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip
L, dev = _hip.lib(), torch.device('cuda:0')
for B, S in ((32, 416), (64, 416), (8, 608)):
    x = torch.randn(B, 3, S, S, device=dev)
    w = torch.randn(32, 3, 3, 3, device=dev) * 0.2
    sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    yp = torch.empty(B, S // 2, S // 2, 32, device=dev)
    z = torch.empty(B, S, S, 32, device=dev)
    stats = torch.zeros(_hip.STATS_REPL * 64, dtype=torch.float64, device=dev)
    st = _hip.stream()
    def inf():
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(sc), _hip.ptr(sh), None, _hip.ptr(yp), None, B, S, S, 3, 32, 0, 32, 0.1, st), 'conv0')
    def trn():
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(w), None, None, _hip.ptr(z), None, _hip.ptr(stats), B, S, S, 3, 32, 32, 0, 1.0, st), 'conv0')
    for name, fn in (('inference (pool)', inf), ('training (z + stats)', trn)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('B=%d %dx%d %-22s %.4f ms  %.1f TF/s' % (B, S, S, name, ms, 2.0 * B * S * S * 27 * 32 / ms / 1e9), flush=True)
