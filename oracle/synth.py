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
