#!/usr/bin/env python
"""A/B of the fused Winograd kernels (Y2_WF_VARIANT: -1 = first generation, 0..3 = feature mask of wino_fused2_kernel, 32 = the
third generation (wino_fused3_kernel: two workgroups per CU); 100 / 136 = Y2_ALGO_WINOGRAD_IMPLICIT on the second / third generation:
the same kernels with the input transform in their loader, no wino_input_kernel; 200 = three-kernel Winograd, 300 = direct) on the
Darknet-19 layer shapes that run it: time per launch (HIP events, best of 3 x reps) and bit-exactness against variant -1.

    python tools/wf_bench.py [--batch 32] [--variants -1,0,1,...] [--reps 10] [--pool] [--stats]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402

SHAPES = [(208, 32, 64), (104, 64, 128), (52, 128, 256), (26, 256, 512), (26, 512, 512), (13, 512, 1024),       # (H = W, Cin, Cout): fprop ...
          (104, 128, 64), (52, 256, 128), (26, 512, 256), (208, 64, 32)]                          # ... and data-gradient shapes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--variants', default='-1,0,1,2,3,4,5,6,7,8,9,13,15')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--pool', action='store_true')
    ap.add_argument('--stats', action='store_true')
    ap.add_argument('--shapes', default='')
    ap.add_argument('--stamps', action='store_true', help='builds with Y2_EXTRA_FLAGS=-DY2_STAMPS (Y2_LIB=...): print the per-stage cycle stamps of workgroup 0')
    ap.add_argument('--kernel-only', action='store_true', help='print the fused kernel alone (min over reps, event hooks) instead of the whole y2_conv_fwd')
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(',')]
    shapes = SHAPES if not args.shapes else [tuple(int(x) for x in sh.split('x')) for sh in args.shapes.split(',')]
    dev = torch.device('cuda:0')
    L, st = _hip.lib(), _hip.stream()
    B = args.batch
    print('%-18s' % 'shape' + ''.join('%9s' % ('v%d' % v) for v in variants) + '   (ms per launch; * = differs from v-1)')
    for H, cin, cout in shapes:
        g = torch.Generator().manual_seed(H + cin)
        x = torch.randn(B, H, H, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
        scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
        shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
        wp = torch.empty(w.numel(), device=dev)
        _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wp), cout, cin, 3, 0, st), 'pack')
        u = _hip.wino_weight(wp, cout, cin)
        y = torch.empty(B, H, H, cout, device=dev)
        yp = torch.empty(B, H // 2, H // 2, cout, device=dev) if (args.pool and H % 2 == 0) else None
        stats = torch.zeros(_hip.STATS_REPL * 2 * cout, dtype=torch.float64, device=dev) if args.stats else None
        p = _hip.ConvParams()
        p.x, p.w, p.scale, p.shift, p.y = x.data_ptr(), u.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr()
        p.y_pool = yp.data_ptr() if yp is not None else None
        p.stats = stats.data_ptr() if stats is not None else None
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.ldp, p.slope, p.algo = B, H, H, cin, cin, cout, 3, cout, cout, 0.1, 2
        p.algo, p.tile = 1, 5
        need = L.y2_conv_fwd_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(need // 4 + 4, device=dev)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        ref = None
        row = '%-18s' % ('%dx%d %d->%d' % (H, H, cin, cout))
        diffs = []
        for v in variants:
            os.environ['Y2_WF_VARIANT'] = str(3 if v == 100 else (32 if v == 136 else v))      # 136: third-generation kernel, implicit
            p.algo = 3 if v in (100, 136) else (1 if v == 200 else (0 if v == 300 else 2))      # 200: three-kernel Winograd (64x128 GEMM tiles), 300: direct
            p.tile = 5 if v == 200 else 0
            p.w = wp.data_ptr() if v == 300 else u.data_ptr()
            y.fill_(float('nan'))
            if yp is not None:
                yp.fill_(float('nan'))
            if stats is not None:
                stats.zero_()
            rc = L.y2_conv_fwd(ctypes.byref(p), st)
            if rc != 0:
                row += '%9s' % ('rc%d' % rc)
                continue
            torch.cuda.synchronize()
            if args.stamps:
                T = B * ((H + 1) // 2) ** 2
                a256 = lambda n: (n + 255) // 256 * 256
                off = (0 if v in (100, 136) else a256(16 * T * cin * 4)) + a256((T + 63) * 4) + 1024
                raw = ws.view(torch.uint8)[off:off + 8000].cpu().numpy().view('uint64')
                d = [int(raw[i + 1]) - int(raw[i]) if raw[i + 1] and raw[i] else 0 for i in range(0, 160)]
                nst = 4 * (cin // 32)
                print('v%d %dx%d %d->%d: stages/tile %d; stamp deltas (cycles):' % (v, H, H, cin, cout, nst))
                per = nst + 2
                for r0 in range(0, 160 - per, per):
                    print('   ', ' '.join('%5d' % x for x in d[r0:r0 + per]))
            out = (y.clone(), yp.clone() if yp is not None else None, stats.clone() if stats is not None else None)
            if v == -1 or ref is None:
                ref = out
            same = torch.equal(out[0], ref[0]) and (yp is None or torch.equal(out[1], ref[1]))
            if stats is not None:       # atomics: order differs run to run, compare to rounding
                same = same and bool(((out[2] - ref[2]).abs() <= 1e-9 * ref[2].abs().max()).all())
            best = float('inf')
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    L.y2_conv_fwd(ctypes.byref(p), st)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.reps)
            if args.kernel_only:          # the GEMM + output-transform kernel alone, from the library's per-kernel event hooks
                L.y2_prof_enable(1)
                for _ in range(args.reps):
                    L.y2_conv_fwd(ctypes.byref(p), st)
                torch.cuda.synchronize()
                L.y2_prof_enable(0)
                nm = ctypes.create_string_buffer(96)
                ms, fl = ctypes.c_float(), ctypes.c_double()
                tk = []
                for i in range(L.y2_prof_count()):
                    L.y2_prof_get(i, nm, 96, ctypes.byref(ms), ctypes.byref(fl))
                    if nm.value.decode().startswith('wino_fused'):
                        tk.append(ms.value)
                best = min(tk)
            row += '%8.4f%s' % (best, ' ' if same else '*')
            if not same:
                diffs.append('v%d: max|d|/rms %.2e' % (v, ((out[0] - ref[0]).abs().max() / ref[0].pow(2).mean().sqrt()).item()))
        print(row + ('   ' + '; '.join(diffs) if diffs else ''), flush=True)
    print('(time includes wino_input_kernel; executed GFLOP per launch = 2*16*T*Cin*Cout)')


if __name__ == '__main__':
    main()
