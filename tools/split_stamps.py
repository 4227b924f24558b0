#!/usr/bin/env python
"""Per-stage cycle stamps of gemm_split_kernel (workgroup 0, wave 0) from a -DY2_STAMPS build of the library:

    Y2_EXTRA_FLAGS=-DY2_STAMPS Y2_OUT=$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip_stamps.so Y2_OBJ=$PWD/yolo2-pytorch_amd/csrc/build_stamps bash yolo2-pytorch_amd/csrc/build.sh
    Y2_LIB=$PWD/yolo2-pytorch_amd/csrc/libyolo2_hip_stamps.so python tools/split_stamps.py [M N K groups]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import _hip

M, N, K, G = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (1568, 1024, 1024, 16)
dev = torch.device('cuda:0')
L, st = _hip.lib(), _hip.stream()
g = torch.Generator().manual_seed(0)
F16 = os.environ.get('STAMPS_F16', '0') == '1'
A = _hip.split_planes(torch.randn(G, M, K, generator=g).to(dev) * (1 / 256.0 if F16 else 1.0), 'f16' if F16 else 'bf16')     # (split_planes scales by 256 in the f16 mode)
B = _hip.split_planes((torch.randn(G, N, K, generator=g) * 0.05).to(dev), 'f16' if F16 else 'bf16')
C = torch.empty(G, M, N, device=dev)
stamps = torch.zeros(256, dtype=torch.int64, device=dev)
os.environ['Y2_GS_STAMPS_PTR'] = str(stamps.data_ptr())
for bk, nw in ((('32', '8'), ('32', '4')) if F16 else (('32', '8'), ('32', '4'), ('16', '4'))):
    os.environ['Y2_SPLIT_BK'] = bk
    os.environ['Y2_SPLIT_WAVES'] = nw
    for _ in range(3):
        _hip.check(L.y2_gemm_split_f16(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), M, N, K, N, G, 1.0, st) if F16 else L.y2_gemm_split(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), M, N, K, N, G, st), 'gemm')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _hip.check(L.y2_gemm_split_f16(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), M, N, K, N, G, 1.0, st) if F16 else L.y2_gemm_split(_hip.ptr(A), _hip.ptr(B), _hip.ptr(C), M, N, K, N, G, st), 'gemm')
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    raw = stamps.cpu().tolist()
    cyc, rt = raw[254] - raw[252], raw[255] - raw[253]
    print('   K loop of workgroup 0: %d shader cycles in %d ticks of the 100 MHz real-time counter = %.2f us -> %.2f GHz; kernel %.1f us' % (cyc, rt, rt / 100.0, cyc / max(rt, 1) / 10.0, ms * 1e3))
    s = stamps.cpu().view(-1, 4).tolist()
    nst = min(K // int(bk) - 2, 62)
    print('BK=%s waves=%s: %.3f ms per launch = %.0f TFLOP/s fp32-equivalent; stamps of workgroup 0 (cycles): wait-DMA / barrier / MFMAs+issue / loop' % (bk, nw, ms, 2.0 * M * N * K * G / ms / 1e9))
    rows = []
    for i in range(nst):
        nxt = s[i + 1][0] if i + 1 < nst else s[i][3]
        rows.append((s[i][1] - s[i][0], s[i][2] - s[i][1], s[i][3] - s[i][2], nxt - s[i][3]))
    for i in list(range(min(6, nst))) + list(range(max(6, nst - 3), nst)):
        print('   stage %2d: %5d %5d %5d %5d' % ((i,) + rows[i]))
    if nst > 4:
        mid = rows[2:-1]
        print('   mean of stages 2..%d: wait %.0f, barrier %.0f, compute %.0f, loop %.0f  (MFMA floor per stage: %d)' %
              (nst - 2, sum(r[0] for r in mid) / len(mid), sum(r[1] for r in mid) / len(mid), sum(r[2] for r in mid) / len(mid), sum(r[3] for r in mid) / len(mid), (24 if F16 else 48) * 32 * int(bk) // 32 * 4 // int(nw)))
