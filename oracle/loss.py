"""torch-CPU restatement of the region loss (model/__init__.py:59-167) with
torch-0.3 mask semantics: `prod(a<b)` is a ByteTensor
mask (:80,91), uint8 logical ops (:92,145), and a `[...,1]` mask selecting from a
`[...,2]`/`[...,C]` tensor broadcasts like masked_select (:154,155,160,162).
Differentiable (autograd) so tests can take d(loss)/d(feature) as ground truth
for the fused HIP loss kernel.
"""
import numpy as np
import torch
import torch.nn.functional as F


def batch_iou_matrix(yx_min1, yx_max1, yx_min2, yx_max2, min=float(np.finfo(np.float32).eps)):
    """utils/iou/torch.py:116-153 in broadcasting form (same operation order)."""
    mn = torch.max(yx_min1.unsqueeze(-2), yx_min2.unsqueeze(-3))
    mx = torch.min(yx_max1.unsqueeze(-2), yx_max2.unsqueeze(-3))
    size = torch.clamp(mx - mn, min=0)
    inter = size[..., 0] * size[..., 1]
    area1 = torch.prod(yx_max1 - yx_min1, -1).unsqueeze(-1)
    area2 = torch.prod(yx_max2 - yx_min2, -1).unsqueeze(-2)
    union = torch.clamp(area1 + area2 - inter, min=min)
    return inter / union


def iou_match(yx_min, yx_max, data):
    """model/__init__.py:59-73: best-IoU GT per (cell, anchor) slot; first max on ties."""
    B, cells, A, _ = yx_min.shape
    m = batch_iou_matrix(yx_min.reshape(B, -1, 2), yx_max.reshape(B, -1, 2), data['yx_min'], data['yx_max'])
    m = m.view(B, cells, A, -1)
    iou, index = m.max(-1)
    flat = index.view(B, -1)
    _data = {}
    for key in ('yx_min', 'yx_max', 'cls'):
        t = data[key]
        if t.dim() == 2:
            g = torch.gather(t, 1, flat).view(B, cells, A)
        else:
            g = torch.gather(t, 1, flat.unsqueeze(-1).expand(-1, -1, t.shape[-1])).view(B, cells, A, -1)
        _data[key] = g
    return m, iou, index, _data


def fit_positive(rows, cols, yx_min, yx_max, anchors):
    """model/__init__.py:76-95: one positive slot per valid GT: cell of its centre, best-shape anchor."""
    B, N, _ = yx_min.shape
    A = anchors.shape[0]
    valid = (yx_min < yx_max).all(-1)  # :80
    center = (yx_min + yx_max) / 2
    ij = torch.floor(center).long()
    index = ij[..., 0] * cols + ij[..., 1]  # :84
    anchors2 = anchors / 2
    m = batch_iou_matrix((yx_min - center).view(1, -1, 2), (yx_max - center).view(1, -1, 2),
                         (-anchors2).view(1, -1, 2), anchors2.view(1, -1, 2)).view(B, N, A)  # :86
    index_anchor = m.max(-1)[1]
    positive = torch.zeros(B, rows * cols, A, dtype=torch.bool)
    for b in range(B):  # :90-94
        v = valid[b]
        positive[b, index[b][v], index_anchor[b][v]] = True
    return positive


def fill_norm(yx_min, yx_max, anchors):
    """model/__init__.py:98-103."""
    center = (yx_min + yx_max) / 2
    ij = torch.floor(center)
    return center - ij, torch.log((yx_max - yx_min) / anchors.view(1, -1, 2))


def loss(anchors, data, pred, threshold):
    """model/__init__.py:138-167.  data: yx_min,yx_max [B,N,2] in CELL units (train.py:57-62 already
    applied), cls int64 [B,N] or one-hot fp32 [B,N,C].  Returns (loss dict, debug dict)."""
    iou = pred['iou']
    rows, cols = pred['feature'].shape[-2:]
    _, _iou, _, _data = iou_match(pred['yx_min'].detach(), pred['yx_max'].detach(), data)  # .data, :142
    positive = fit_positive(rows, cols, data['yx_min'], data['yx_max'], anchors)
    negative = ~positive & (_iou < threshold)  # :145
    _center_offset, _size_norm = fill_norm(_data['yx_min'], _data['yx_max'], anchors)
    _cls = _data['cls']
    sq = lambda t: t * t
    out = {}
    out['foreground'] = sq(iou[positive] - _iou[positive]).sum()  # :151
    out['background'] = sq(iou[negative]).sum()  # :152
    out['center'] = sq(pred['center_offset'][positive] - _center_offset[positive]).sum()  # :154
    out['size'] = sq(pred['size_norm'][positive] - _size_norm[positive]).sum()  # :155
    if 'logits' in pred:
        logits = pred['logits']
        if _cls.dim() > 3:
            out['cls'] = sq(F.softmax(logits, -1)[positive] - _cls[positive]).sum()  # :160
        else:
            out['cls'] = F.cross_entropy(logits[positive].view(-1, logits.shape[-1]), _cls[positive].view(-1))  # :162 (mean)
    cnt = float(np.multiply.reduce(positive.shape))  # :164
    for key in out:
        out[key] = out[key] / cnt
    return out, dict(iou=_iou, data=_data, positive=positive, negative=negative)


HPARAM = dict(foreground=5.0, background=1.0, center=1.0, size=1.0, cls=1.0)  # config.ini:100-105


def total(loss_dict, hparam=HPARAM):
    """train.py:348-349."""
    return sum(loss_dict[k] * hparam[k] for k in loss_dict)
