#!/usr/bin/env python
"""Direct vs Winograd weight gradient on the Darknet-19 3x3 layer shapes (MI355X tuning aid).

    python tools/wgrad_bench.py [--batch 64] [--size 416] [--reps 5]

TF/s are direct-equivalent (2*9*Cin*Cout*B*H*W)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402

LAYERS = [('l1.4', 64, 128, 4), ('l1.8', 128, 256, 8), ('l1.12', 256, 512, 16), ('l2.1', 512, 1024, 32), ('l2.6', 1024, 1024, 32), ('l3.0', 1280, 1024, 32)]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    L = _hip.lib()
    dev = torch.device('cuda:0')
    B = args.batch
    for name, cin, cout, div in LAYERS:
        H = W = args.size // div
        x = torch.randn(B, H, W, cin, device=dev)
        dz = torch.randn(B, H, W, cout, device=dev)
        dw = torch.zeros(cout * 9 * cin, device=dev)
        ws = torch.empty(L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout) // 4 + 4, device=dev)
        st = _hip.stream()
        flops = 2.0 * 9 * cin * cout * B * H * W

        def direct():
            dw.zero_()
            _hip.check(L.y2_conv_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dw), B, H, W, cin, cin, cout, cout, 3, st), 'wgrad')

        def wino():
            _hip.check(L.y2_wino_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dw), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, st), 'wino_wgrad')
        def wino6():
            _hip.check(L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dw), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, 2, st), 'wino6_wgrad')
        td, tw, t6 = timeit(direct, args.reps), timeit(wino, args.reps), timeit(wino6, args.reps)
        print('%-6s B=%d %3dx%-3d %4d->%-4d direct %7.3f ms | F(3,2) %7.3f ms (x + dz transformed here) | F(3,4) %7.3f ms' %
              (name, B, H, W, cin, cout, td, tw, t6), flush=True)


if __name__ == '__main__':
    main()
