"""Synthetic workloads of bench.py and tools/ (SURVEY.md 8d): seeded inputs, loss weights and random-init weights.

Neutral ground between the product and the oracle: the GPU legs of bench.py take their inputs from HERE and never import
`oracle/` (the oracle stays the checker and the cpu_baseline leg).  Everything is a deterministic function of a seed.
"""
import configparser
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
APP = os.path.join(ROOT, 'yolo2-pytorch_amd')
for _p in (ROOT, APP):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# (height, width) rows in cell units: config/anchors/voc.tsv with the columns swapped as utils.get_anchors does (utils/__init__.py:78-81)
ANCHORS_VOC = np.array([[1.19, 1.08], [4.41, 3.42], [11.38, 6.63], [5.11, 9.42], [10.52, 16.62]], np.float32)
# config.ini:100-105 ([hparam]): loss_total = 5 * foreground + background + center + size + cls (train.py:348-349)
HPARAM = dict(foreground=5.0, background=1.0, center=1.0, size=1.0, cls=1.0)
THRESHOLD = 0.6     # [model] threshold: IoU below which an unmatched slot counts as background (config.ini, model/__init__.py:147)


def images(B, S, seed=1):
    """The reference's own checksum convention: torch.randn(B,3,S,S) (checksum_torch.py:55)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, S, S, generator=g)


def labels(B, S, C=20, nmax=8, seed=2):
    """Contract of utils/data.py:29-42,114-133: per image 1..nmax boxes (centre ~U(0.05,0.95)*S, size ~U(0.05,0.6)*S, clipped),
    zero-padded [B,nmax,2] pixel boxes + int64 class ids."""
    rng = np.random.RandomState(seed)
    yx_min = np.zeros((B, nmax, 2), np.float32)
    yx_max = np.zeros((B, nmax, 2), np.float32)
    cls = np.zeros((B, nmax), np.int64)
    for b in range(B):
        n = rng.randint(1, nmax + 1)
        c = rng.uniform(0.05, 0.95, (n, 2)) * S
        s = rng.uniform(0.05, 0.6, (n, 2)) * S
        yx_min[b, :n] = np.clip(c - s / 2, 0, S)
        yx_max[b, :n] = np.clip(c + s / 2, 0, S)
        cls[b, :n] = rng.randint(0, C, n)
    return dict(yx_min=torch.from_numpy(yx_min), yx_max=torch.from_numpy(yx_max), cls=torch.from_numpy(cls))


def randomize(dnn, seed=0, head_scale=1 / 40.0, gamma=(0.5, 1.0)):
    """SURVEY.md 8d weights: the plugin's own init (kaiming conv, gamma 1, beta 0 - model/yolo2.py:117-123) under `seed`, then
    randomised BatchNorm parameters / buffers so that folding is exercised, and the head scaled so that exp(size_norm) stays
    finite on randn images (feature rms ~ 1)."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in dnn.modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * gamma[1] + gamma[0])
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
        head = [m for m in dnn.modules() if isinstance(m, nn.Conv2d)][-1]
        head.weight.mul_(head_scale)
        if head.bias is not None:
            head.bias.mul_(head_scale)
    return dnn


def build_model(num_cls, dev, arch='darknet', seed=0):
    """(model.Inference in eval mode on `dev`, anchors) for `arch` in {'darknet', 'tiny', 'resnet18'...'resnet152'}."""
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}, 'model': {'pretrained': '0'}})
    anchors = torch.from_numpy(ANCHORS_VOC)
    torch.manual_seed(seed)
    if arch == 'darknet':
        dnn = model.yolo2.Darknet(model.ConfigChannels(cfg), anchors, num_cls)
        randomize(dnn, seed, 1 / 40.0)
    elif arch == 'tiny':
        dnn = model.yolo2.Tiny(model.ConfigChannels(cfg), anchors, num_cls)
        randomize(dnn, seed, 1 / 40.0)
    else:
        import model.resnet
        dnn = getattr(model.resnet, arch)(model.ConfigChannels(cfg), anchors, num_cls)
        randomize(dnn, seed, 0.25, gamma=(0.25, 0.5))      # smaller BN gains keep the residual sums of 16-50 blocks in range
    return model.Inference(cfg, dnn, anchors).to(dev).eval(), anchors
