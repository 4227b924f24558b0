"""`model` — detection-head math of YOLOv2, MI355X-native mirror of the reference's model/__init__.py.

Same names and call signatures as the reference (model/__init__.py:29-179):
ConfigChannels, output_channels, meshgrid, Inference, _inference, loss.  The
arithmetic (decode, matching, region loss) runs in libyolo2_hip.so; torch only
provides device memory, streams and autograd bookkeeping.
"""
import logging

import numpy as np
import torch
import torch.nn as nn

import _hip


class ConfigChannels(object):
    """Channel-count oracle of the plugin constructors (model/__init__.py:29-43).  Without a checkpoint every layer gets its
    default width; with one (pruned models, `--finetune`) the width is read off the named tensor - `fn(state_dict[name])`, by
    default its first dimension.  Stateful: `channels` always holds the width decided last (the next layer's input width;
    3 = the image planes before the first call)."""

    def __init__(self, config, state_dict=None, channels=3):
        self.config, self.state_dict, self.channels = config, state_dict, channels

    def __call__(self, default, name, fn=lambda var: var.size(0)):
        width = default
        if self.state_dict is not None:
            width = fn(self.state_dict[name])
            if width != default:
                logging.warning('%s: change number of output channels from %d to %d' % (name, default, width))
        self.channels = width
        return width


def output_channels(num_anchors, num_cls):
    """Head width (model/__init__.py:46-50): 5 box/objectness rows per anchor plus one row per class when there are several."""
    return num_anchors * (5 + (num_cls if num_cls > 1 else 0))


def meshgrid(rows, cols, swap=False):
    """Cell-offset table of the decode (model/__init__.py:53-56): row k is (k // rows, k % rows) - equal to (row, col) of cell k
    only for square grids, a quirk the decode kernel reproduces - or the two columns exchanged with swap=True.  Kept for API
    compatibility; y2_decode generates the offsets itself."""
    k = torch.arange(0, rows * cols)
    fast, slow = (k % rows).view(-1, 1), torch.div(k, rows, rounding_mode='floor').view(-1, 1)
    return torch.cat([fast, slow] if swap else [slow, fast], 1)


_ANCHOR_CACHE = {}


def _device_anchors(anchors, dev):
    """Device copy of the (host) anchor table, cached: a per-call H2D copy from pageable memory is synchronous."""
    if anchors.is_cuda and anchors.dtype == torch.float32 and anchors.is_contiguous():
        return anchors
    key = (anchors.data_ptr(), anchors._version, tuple(anchors.shape), str(dev))
    hit = _ANCHOR_CACHE.get(key)
    if hit is None:
        if len(_ANCHOR_CACHE) > 64:
            _ANCHOR_CACHE.clear()
        hit = (anchors.to(device=dev, dtype=torch.float32).contiguous(), anchors)   # keep the host tensor alive: data_ptr is the key
        _ANCHOR_CACHE[key] = hit
    return hit[0]


def decode(feature_nhwc, anchors, num_anchors, want_prob=False):
    """y2_decode on a contiguous [B, rows, cols, A*(5+C)] head image.  Returns dict of fp32 GPU tensors."""
    _hip.require_gpu(feature_nhwc)
    L = _hip.lib()
    B, rows, cols, ch = feature_nhwc.shape
    A = num_anchors
    E = ch // A
    C = E - 5
    cells = rows * cols
    dev = feature_nhwc.device
    anchors = _device_anchors(anchors, dev)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = dict(iou=new(B, cells, A), center_offset=new(B, cells, A, 2), size_norm=new(B, cells, A, 2),
               yx_min=new(B, cells, A, 2), yx_max=new(B, cells, A, 2))
    if want_prob:
        out['prob'] = new(B, cells, A, max(C, 1))
        out['prob_cls'] = new(B, cells, A)
        out['cls'] = torch.empty(B, cells, A, dtype=torch.int32, device=dev)
        if C == 0:
            out['prob'].fill_(1.0)  # detect.get_logits: ones when single-class (detect.py:43-48)
    _hip.check(L.y2_decode(_hip.ptr(feature_nhwc), _hip.ptr(anchors), B, rows, cols, A, C,
                           _hip.ptr(out['iou']), _hip.ptr(out['center_offset']), _hip.ptr(out['size_norm']),
                           _hip.ptr(out['yx_min']), _hip.ptr(out['yx_max']),
                           _hip.ptr(out['prob']) if want_prob and C > 0 else None,
                           _hip.ptr(out.get('prob_cls')), _hip.ptr(out.get('cls')), _hip.stream()), 'y2_decode')
    return out


class Inference(nn.Module):
    """model/__init__.py:110-135.  `dnn` is any plugin with the reference's forward contract."""

    def __init__(self, config, dnn, anchors):
        nn.Module.__init__(self)
        self.config = config
        self.dnn = dnn
        self.anchors = anchors

    def forward(self, x):
        feature = self.dnn(x)
        if torch.is_grad_enabled() and feature.requires_grad:
            from model import train_graph
            return train_graph.inference_forward(self, feature)
        A = self.anchors.size(0)
        _feature = feature.permute(0, 2, 3, 1).contiguous()  # free when the plugin produced NHWC memory
        d = decode(_feature, self.anchors, A)
        B, rows, cols, ch = _feature.shape
        _f = _feature.view(B, rows * cols, A, -1)
        logits = _f[:, :, :, 5:] if _f.size(-1) > 5 else None
        return feature, d['iou'], d['center_offset'], d['size_norm'], d['yx_min'], d['yx_max'], logits


def loss(anchors, data, pred, threshold):
    """model/__init__.py:138-167 — fused HIP region loss (matching + masks + 5 terms)."""
    from model import train_graph
    return train_graph.loss(anchors, data, pred, threshold)


def weighted_total(loss_, hparam):
    """The weighted sum of the loss terms, train.py:348-349: `sum(loss[key] * hparam[key] for key in loss)`."""
    from model import train_graph
    return train_graph.weighted_total(loss_, hparam)


_PRED_KEYS = ('feature', 'iou', 'center_offset', 'size_norm', 'yx_min', 'yx_max', 'logits')


def _inference(inference, tensor):
    """The head's 7-tuple as the dict train / eval / detect consume (model/__init__.py:170-179); `logits` is absent for
    single-class models."""
    pred = {k: v for k, v in zip(_PRED_KEYS, inference(tensor)) if v is not None}
    if 'logits' in pred:
        pred['logits'] = pred['logits'].contiguous()
    return pred
