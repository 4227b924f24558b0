#!/usr/bin/env python
"""What the fp32 matrix pipe sustains on this box, and whether vector instructions run beside it (tools/probes/mfma_peak.hip).

The roofline's `peak` for the fp32-input MFMA kernels is 157.3 TFLOP/s (256 CUs x 256 flop/clk x 2.4 GHz, MI355X_MICROARCH.md).  This prints what a
register-only loop of v_mfma_f32_32x32x2_f32 reaches at 1 / 2 / 4 waves per SIMD (the clock the chip holds under matrix load is part of it), and the
same loop with 2 ... 16 independent v_fma_f32 behind every MFMA: if the vector instructions ran in the MFMA's shadow the time would not move up to 16."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libmfma_peak.so'))
lib.mfma_peak_run.restype = ctypes.c_float
lib.mfma_peak_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256, device='cuda:0')
ITERS = 20000
print('waves/SIMD  valu/mfma   ms      TFLOP/s (fp32 MFMA)   of 157.3   ns per MFMA per SIMD (64 clk at 2.4 GHz = 26.7)')
for wps in (1, 2, 4):
    for v in (0, 2, 4, 8, 12, 16):
        blocks = 256 * wps
        ms = lib.mfma_peak_run(v, blocks, ITERS, ctypes.c_void_p(out.data_ptr()))
        mf = blocks * 4 * ITERS * 4                      # MFMAs issued
        tf = mf * 4096 / ms / 1e9
        print('%6d %10d %9.3f %12.1f %16.3f %14.1f' % (wps, v, ms, tf, tf / 157.3, ms * 1e6 / (ITERS * 4 * wps)))
