#!/usr/bin/env python
"""Per-layer table of the Darknet-19 detect plan as autotuned on this GPU: algorithm (direct / Winograd), tile, time, rates.

    python tools/layer_table.py [--batch 32] [--size 416] [--reps 10]

Each of the 22 y2_conv_fwd problems of the plan is launched alone `reps` times between HIP events.  TF/s (direct-equivalent)
= 2*Cin*Cout*k*k*B*H*W / t; TF/s (executed) counts 16 products per 2x2 tile for Winograd layers."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402
import bench_data  # noqa: E402

TILES = {0: 'auto', 1: '128x128', 2: '128x64', 3: '64x64', 5: '64x128', 6: '128x32'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    inf, anchors = bench_data.build_model(20, dev, 'darknet')
    dnn = inf.dnn
    x = bench_data.images(args.batch, args.size, seed=1).to(dev)
    with torch.no_grad():
        for _ in range(2):
            dnn.forward_nhwc(x)
    plan = dnn._plan_cache[1]
    L, st = _hip.lib(), _hip.stream()
    names = [n for n, _, _ in dnn._blocks()[0][1:]] + ['passthrough'] + [n for n, _, _ in dnn._blocks()[1]] + [n for n, _, _ in dnn._blocks()[2]]
    tot = 0.0
    print('%-12s %5s %5s %4s %2s %-9s %-8s %9s %8s %8s' % ('layer', 'Cin', 'Cout', 'HxW', 'k', 'algorithm', 'tile', 'ms', 'TF/s eq', 'TF/s ex'))
    for i in range(plan['n']):
        p = plan['arr'][i]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.y2_conv_fwd(ctypes.byref(p), st)
        e0.record()
        for _ in range(args.reps):
            L.y2_conv_fwd(ctypes.byref(p), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        tot += ms
        eq = 2.0 * p.Cin * p.Cout * p.ksize ** 2 * p.B * p.H * p.W
        ex = 2.0 * p.Cin * p.Cout * 16 * p.B * ((p.H + 1) // 2) * ((p.W + 1) // 2) if p.algo in (1, 2, 3) else eq
        print('%-12s %5d %5d %4d %2d %-9s %-8s %9.4f %8.1f %8.1f' % (names[i] if i < len(names) else '?', p.Cin, p.Cout, p.H, p.ksize,
              {1: 'winograd', 2: 'wino-fused', 3: 'wino-impl'}.get(p.algo, 'direct'), TILES.get(p.tile, str(p.tile)), ms, eq / ms / 1e9, ex / ms / 1e9), flush=True)
    print('sum of the 22 layers: %.4f ms (B=%d)' % (tot, args.batch))


if __name__ == '__main__':
    main()
