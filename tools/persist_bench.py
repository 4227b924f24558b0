#!/usr/bin/env python
"""1x1 convolutions (short K loops): the one-tile-per-workgroup tiles (5 / 3 / 2 / 1) against their persistent forms (15 / 13 / 12 / 11) on the Darknet-19 and
ResNet-50 1x1 shapes.  Prints ms and TF/s per tile id, best of each family."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch

import _hip

L = _hip.lib()
d = torch.device('cuda', 0)
SHAPES = [  # B, H, W, Cin, Cout
    (64, 104, 104, 128, 64), (64, 52, 52, 256, 128), (64, 26, 26, 512, 256), (64, 13, 13, 1024, 512), (64, 26, 26, 512, 64),
    (64, 104, 104, 64, 128), (64, 52, 52, 128, 256), (64, 26, 26, 256, 512), (64, 13, 13, 512, 1024),     # data gradients of the same layers
    (32, 104, 104, 128, 64), (32, 52, 52, 256, 128), (32, 26, 26, 512, 256), (32, 13, 13, 1024, 512),
    (32, 152, 152, 64, 64), (32, 152, 152, 64, 256), (32, 152, 152, 256, 64), (32, 76, 76, 512, 128), (32, 76, 76, 128, 512), (32, 38, 38, 1024, 256), (32, 38, 38, 256, 1024),
    (32, 19, 19, 2048, 512), (32, 19, 19, 512, 2048),
]
for B, H, W, cin, cout in SHAPES:
    x = torch.randn(B, H, W, cin, device=d)
    w = torch.randn(cout, cin, device=d) * 0.05
    y = torch.empty(B, H, W, cout, device=d)
    row = []
    for tile in (5, 3, 2, 1, 15, 13, 12, 11):
        p = _hip.ConvParams()
        p.x, p.w, p.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope, p.tile = B, H, W, cin, cin, cout, 1, cout, 1.0, tile
        if _hip.conv_workspace(p, d) < 0 or L.y2_conv_fwd(ctypes.byref(p), _hip.stream()) != 0:
            row.append((tile, None))
            continue
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.y2_conv_fwd(ctypes.byref(p), _hip.stream())
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row.append((tile, best))
    fl = 2.0 * B * H * W * cin * cout
    old = min((t for tl, t in row[:4] if t), default=None)
    new = min((t for tl, t in row[4:] if t), default=None)
    print('B=%d %3dx%-3d %4d->%-4d  ' % (B, H, W, cin, cout) + '  '.join('%d:%s' % (tl, 'n/a' if t is None else '%.3f' % t) for tl, t in row)
          + '   best %.3f ms (%.0f TF/s) -> %s' % (old, fl / old / 1e9, 'n/a' if new is None else '%.3f ms (%.0f TF/s) %+.0f%%' % (new, fl / new / 1e9, (old / new - 1) * 100)), flush=True)
