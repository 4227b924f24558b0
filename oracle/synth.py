"""Seeded synthetic inputs shared by tests, smoke() and bench.py."""
import numpy as np
import torch

ANCHORS_VOC = np.array([[1.19, 1.08], [4.41, 3.42], [11.38, 6.63], [5.11, 9.42], [10.52, 16.62]], np.float32)  # (h, w): config/anchors/voc.tsv columns swapped, utils/__init__.py:78-81


def images(B, S, seed=1, kind='randn'):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, S, S, generator=g) if kind == 'randn' else torch.rand(B, 3, S, S, generator=g)


def labels(B, S, C=20, nmax=8, seed=2, onehot=False):
    """utils/data.py:29-42 contract: zero-padded [B,Nmax,2] pixel boxes + cls."""
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
    data = dict(yx_min=torch.from_numpy(yx_min), yx_max=torch.from_numpy(yx_max), cls=torch.from_numpy(cls))
    if onehot:
        oh = torch.zeros(B, nmax, C)
        oh.scatter_(2, data['cls'].unsqueeze(-1), 1.0)
        data['cls'] = oh
    return data


def edge_labels(S_h, S_w, C=20, onehot=False):
    """[B=5, Nmax=6] pixel boxes (yx_min, yx_max) + cls, zero padded."""
    boxes = [
        [],                                                                                   # 0: no object
        [(100, 120, 220, 260, 3), (100, 120, 220, 260, 3)],                                   # 1: identical twice
        [(96, 96, 160, 160, 1), (64, 48, 192, 208, 7), (120, 124, 136, 132, 2)],              # 2: three boxes, one centre cell
        [(S_h - 40, S_w - 60, S_h, S_w, 5), (0, 0, 30, 50, 6), (S_h - 1, S_w - 1, S_h, S_w, 8)],  # 3: last cell, first cell, 1-px box
        [(50, 60, 50, 200, 4), (10, 300, 150, 400, 9), (0, 0, S_h, S_w, 0), (200, 10, 380, 90, 19),
         (210, 20, 370, 80, 19), (33, 33, 34, 34, 11)],                                       # 4: degenerate first, whole image, near-duplicates
    ]
    B, nmax = len(boxes), 6
    yx_min = np.zeros((B, nmax, 2), np.float32)
    yx_max = np.zeros((B, nmax, 2), np.float32)
    cls = np.zeros((B, nmax), np.int64)
    for b, lst in enumerate(boxes):
        for i, (y0, x0, y1, x1, c) in enumerate(lst):
            yx_min[b, i] = (y0, x0)
            yx_max[b, i] = (min(y1, S_h), min(x1, S_w))
            cls[b, i] = c % C
    data = dict(yx_min=torch.from_numpy(yx_min), yx_max=torch.from_numpy(yx_max), cls=torch.from_numpy(cls))
    if onehot:
        oh = torch.zeros(B, nmax, C)
        oh.scatter_(2, data['cls'].unsqueeze(-1), 1.0)
        # padded rows carry an all-zero class vector in the reference's data pipeline (utils/data.py padding)
        valid = (data['yx_min'] < data['yx_max']).all(-1, keepdim=True).float()
        data['cls'] = oh * valid
    return data


def edge_feature(B, A, C, rows, cols, seed=23):
    g = torch.Generator().manual_seed(seed)
    return 0.5 * torch.randn(B, A * (5 + C), rows, cols, generator=g)


EDGE_CASES = (('sq13', 416, 13), ('sq19', 608, 19), ('sq10', 320, 10))


def norm_data(data, height, width, rows, cols):
    """train.py:57-62: GT pixels -> cell units."""
    scale = torch.tensor([rows / height, cols / width], dtype=torch.float32).view(1, 1, 2)
    out = dict(data)
    out['yx_min'] = data['yx_min'] * scale
    out['yx_max'] = data['yx_max'] * scale
    return out


def nms_boxes(n, seed=3, grid=13.0):
    rng = np.random.RandomState(seed + n)
    c = rng.uniform(0, grid, (n, 2)).astype(np.float32)
    s = rng.uniform(0.5, 6.5, (n, 2)).astype(np.float32)
    score = (rng.permutation(n).astype(np.float32) + 1) / np.float32(n + 1)
    return score, (c - s / 2).astype(np.float32), (c + s / 2).astype(np.float32)
