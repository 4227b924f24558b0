#!/usr/bin/env python
"""Per-layer sweep of y2_conv_fwd over the Darknet-19 layer shapes x tile configs (MI355X tuning aid).

    python tools/layer_bench.py [--batch 32] [--size 416] [--tiles 0,1,2,3,5] [--reps 5]

Prints one line per (layer, tile): ms, TFLOP/s, fraction of the 157.3 TF fp32-MFMA peak.  Random data (never zeros:
DVFS inflates zero-filled numbers, cdna_hip_programming.md rule 25)."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402

# (name, Cin, Cout, k, spatial divisor, pool_out)
LAYERS = [('l1.2', 32, 64, 3, 2, True), ('l1.4', 64, 128, 3, 4, False), ('l1.5', 128, 64, 1, 4, False), ('l1.6', 64, 128, 3, 4, True),
          ('l1.8', 128, 256, 3, 8, False), ('l1.9', 256, 128, 1, 8, False), ('l1.10', 128, 256, 3, 8, True),
          ('l1.12', 256, 512, 3, 16, False), ('l1.13', 512, 256, 1, 16, False), ('l1.16', 256, 512, 3, 16, True),
          ('pass', 512, 64, 1, 16, False),
          ('l2.1', 512, 1024, 3, 32, False), ('l2.2', 1024, 512, 1, 32, False), ('l2.6', 1024, 1024, 3, 32, False),
          ('l3.0', 1280, 1024, 3, 32, False), ('head', 1024, 125, 1, 32, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--tiles', default='0,1,2,3,5')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--no-split', action='store_true')
    ap.add_argument('--wino', type=int, default=0, help='3x3 layers through algo 1 (Winograd F(2x2,3x3)) or 2 (fused GEMM + output transform); TF/s stay direct-equivalent')
    args = ap.parse_args()
    L = _hip.lib()
    dev = torch.device('cuda:0')
    B = args.batch
    rows = []
    for name, cin, cout, k, div, pool in LAYERS:
        if args.only and name not in args.only.split(','):
            continue
        H = W = args.size // div
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(cout * k * k * cin, device=dev) * 0.05
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev) * 0.1
        y = torch.empty(B, H, W, cout, device=dev)
        yp = torch.empty(B, H // 2, W // 2, cout, device=dev) if pool else None
        flops = 2.0 * cin * cout * k * k * B * H * W
        for tile in [int(t) for t in args.tiles.split(',')]:
            p = _hip.ConvParams()
            p.x, p.w, p.scale, p.shift = x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr()
            if pool:
                p.y, p.y_pool, p.ldp = None, yp.data_ptr(), cout
            else:
                p.y, p.ldy = y.data_ptr(), cout
            p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, H, W, cin, cin, cout, k
            p.slope, p.tile = 0.1, tile
            if args.wino and k == 3:
                u = torch.empty(16 * cout * cin, device=dev)
                _hip.check(L.y2_wino_weight(w.data_ptr(), u.data_ptr(), cout, cin, _hip.stream()), 'wino_weight')
                p.w, p.algo = u.data_ptr(), args.wino
                _hip.conv_workspace(p, dev)
            elif not args.no_split:
                _hip.conv_workspace(p, dev)
            st = _hip.stream()
            rc = L.y2_conv_fwd(ctypes.byref(p), st)
            if rc != 0:
                print('%-6s tile %d rc %d' % (name, tile, rc))
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                L.y2_conv_fwd(ctypes.byref(p), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            tf = flops / ms / 1e9
            rows.append(dict(layer=name, tile=tile, ms=round(ms, 4), tflops=round(tf, 1), frac=round(tf / 157.3, 3), M=B * H * W, N=cout, K=cin * k * k))
            print('%-6s M=%8d N=%5d K=%6d tile %d : %8.4f ms  %6.1f TF/s  %.3f' % (name, B * H * W, cout, cin * k * k, tile, ms, tf, tf / 157.3), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'layer_bench.json'), 'w'))


if __name__ == '__main__':
    main()
