#!/usr/bin/env python
"""Layer-level A/B of the Winograd algorithms incl. the split-bf16 GEMM (Y2_ALGO_WINOGRAD_SPLIT): ms per layer and the executed
fp32-equivalent TFLOP/s of the whole layer (transforms included) for the deep 3x3 layers of Darknet-19 at 416x416.

    python tools/split_bench.py [--batch 32]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
dev = torch.device('cuda:0')
L, st = _hip.lib(), _hip.stream()
B = args.batch
LAYERS = [('104x104 64->128', 104, 64, 128), ('52x52 128->256', 52, 128, 256), ('26x26 256->512', 26, 256, 512), ('13x13 512->1024', 13, 512, 1024),
          ('13x13 1024->1024', 13, 1024, 1024), ('13x13 1280->1024', 13, 1280, 1024)]
print('%-20s %10s %10s %10s %10s %10s %10s %10s   (ms; executed Winograd TFLOP/s in brackets)' % ('layer', 'direct', 'wino', 'fused', 'split32x8', 'split32x4', 'f16x3 x8', 'f16x3 x4'))
for name, H, cin, cout in LAYERS:
    g = torch.Generator().manual_seed(H + cin)
    x = torch.randn(B, H, H, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    wp = torch.empty(w.numel(), device=dev)
    _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wp), cout, cin, 3, 0, st), 'pack')
    u = _hip.wino_weight(wp, cout, cin)
    us = _hip.split_planes(u, 'bf16')
    uh = _hip.split_planes(u, 'f16')
    y = torch.empty(B, H, H, cout, device=dev)
    scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    exe = 2.0 * cin * cout * 16 * B * ((H + 1) // 2) ** 2
    row = []
    ref = None
    for algo, wt, env in ((0, wp, None), (1, u, None), (2, u, None), (4, us, ('32', '8')), (4, us, ('32', '4')), (5, uh, ('32', '8')), (5, uh, ('32', '4'))):
        if env:
            os.environ['Y2_SPLIT_BK'], os.environ['Y2_SPLIT_WAVES'] = env
        p = _hip.ConvParams()
        p.x, p.w, p.y, p.scale, p.shift = x.data_ptr(), wt.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr()
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope, p.algo, p.tile = B, H, H, cin, cin, cout, 3, cout, 0.1, algo, (5 if algo == 1 else 0)
        if _hip.conv_workspace(p, dev) < 0 or L.y2_conv_fwd(ctypes.byref(p), st) != 0:
            row.append('n/a')
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        err = (y - ref).abs().max().item() / ref.pow(2).mean().sqrt().item()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.y2_conv_fwd(ctypes.byref(p), st)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row.append('%.3f[%3.0f]%s' % (best, (exe if algo else exe * 36 / 16) / best / 1e9, '' if err < 5e-5 else ' ERR %.1e' % err))
    print('%-20s %10s %10s %10s %10s %10s %10s %10s' % tuple([name] + row))
