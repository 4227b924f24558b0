#!/usr/bin/env python
"""max|err| / rms(fp64 reference) of the direct, Winograd (algo 1) and fused Winograd (algo 2) convolution on Darknet-19 layer
shapes (B=2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import test_gpu_kernels as T  # noqa: E402

torch.set_num_threads(32)
for cin, cout, hw in ((64, 128, 104), (128, 256, 52), (256, 512, 26), (512, 1024, 13), (1024, 1024, 13), (1280, 1024, 13)):
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(2, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    out = []
    for wino in (0, 1, 2):
        y = T.run_conv(x, w, None, None, 1.0, 3, wino=wino)['y']
        out.append(T.rel_err(y.permute(0, 3, 1, 2), ref))
    print('%4d->%-4d %2dx%-2d  direct %.2e  winograd %.2e  winograd-fused %.2e' % (cin, cout, hw, hw, out[0], out[1], out[2]), flush=True)
