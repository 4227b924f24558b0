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
    """model/__init__.py:29-43: per-layer channel counts follow a checkpoint when one is given (pruned models)."""

    def __init__(self, config, state_dict=None, channels=3):
        self.config = config
        self.state_dict = state_dict
        self.channels = channels

    def __call__(self, default, name, fn=lambda var: var.size(0)):
        if self.state_dict is None:
            self.channels = default
        else:
            var = self.state_dict[name]
            self.channels = fn(var)
            if self.channels != default:
                logging.warning('%s: change number of output channels from %d to %d' % (name, default, self.channels))
        return self.channels


def output_channels(num_anchors, num_cls):
    """model/__init__.py:46-50."""
    if num_cls > 1:
        return num_anchors * (5 + num_cls)
    else:
        return num_anchors * 5


def meshgrid(rows, cols, swap=False):
    """model/__init__.py:53-56 (kept for API compatibility; the decode kernel generates the grid itself)."""
    i = torch.arange(0, rows).repeat(cols).view(-1, 1)
    j = torch.arange(0, cols).view(-1, 1).repeat(1, rows).view(-1, 1)
    return torch.cat([i, j], 1) if swap else torch.cat([j, i], 1)


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


def _inference(inference, tensor):
    """model/__init__.py:170-179."""
    feature, iou, center_offset, size_norm, yx_min, yx_max, logits = inference(tensor)
    pred = dict(
        feature=feature, iou=iou,
        center_offset=center_offset, size_norm=size_norm,
        yx_min=yx_min, yx_max=yx_max,
    )
    if logits is not None:
        pred['logits'] = logits.contiguous()
    return pred
