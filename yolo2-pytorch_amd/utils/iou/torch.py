"""`utils.iou.torch` — pairwise IoU on (y, x) min/max boxes, MI355X-native (utils/iou/torch.py:24-61,116-153,216-233).

Same function names and argument meaning as the reference; the arithmetic runs in y2_iou_matrix / y2_iou_pair
(one fused kernel, no `repeat`-materialised temporaries) and is bit-identical to the reference's fp32 sequence.
CPU tensors (the reference's unit tests run on them) go to the library's host entry points y2_iou_*_host.
"""
import numpy as np
import torch

import _hip

EPS = float(np.finfo(np.float32).eps)


def _host(ts):
    return [t.detach().to(torch.float32).contiguous() for t in ts]


def _run(yx_min1, yx_max1, yx_min2, yx_max2, batched, min, mode):
    cpu = not yx_min1.is_cuda
    if cpu:      # CPU tensors (the reference's own unit tests, utils/iou/torch.py:64-113): the library's host implementation
        a1, b1, a2, b2 = _host((yx_min1, yx_max1, yx_min2, yx_max2))
    else:
        _hip.require_gpu(yx_min1, yx_max1, yx_min2, yx_max2)
        a1, b1, a2, b2 = (_hip.f32c(t) for t in (yx_min1, yx_max1, yx_min2, yx_max2))
    if batched:
        Bt, N1, N2 = a1.size(0), a1.size(1), a2.size(1)
        out = torch.empty(Bt, N1, N2, dtype=torch.float32, device=a1.device)
    else:
        Bt, N1, N2 = 1, a1.size(0), a2.size(0)
        out = torch.empty(N1, N2, dtype=torch.float32, device=a1.device)
    if cpu:
        _hip.check(_hip.lib().y2_iou_matrix_host(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), Bt, N1, N2, min, mode, out.data_ptr()), 'y2_iou_matrix_host')
        return out
    _hip.check(_hip.lib().y2_iou_matrix(_hip.ptr(a1), _hip.ptr(b1), _hip.ptr(a2), _hip.ptr(b2), Bt, N1, N2, min, mode,
                                        _hip.ptr(out), _hip.stream()), 'y2_iou_matrix')
    return out


def intersection_area(yx_min1, yx_max1, yx_min2, yx_max2):
    return _run(yx_min1, yx_max1, yx_min2, yx_max2, False, EPS, 1)


def iou_matrix(yx_min1, yx_max1, yx_min2, yx_max2, min=EPS):
    return _run(yx_min1, yx_max1, yx_min2, yx_max2, False, min, 0)


def batch_intersection_area(yx_min1, yx_max1, yx_min2, yx_max2):
    return _run(yx_min1, yx_max1, yx_min2, yx_max2, True, EPS, 1)


def batch_iou_matrix(yx_min1, yx_max1, yx_min2, yx_max2, min=EPS):
    return _run(yx_min1, yx_max1, yx_min2, yx_max2, True, min, 0)


def batch_iou_pair(yx_min1, yx_max1, yx_min2, yx_max2, min=EPS):
    if not yx_min1.is_cuda:
        a1, b1, a2, b2 = _host((yx_min1, yx_max1, yx_min2, yx_max2))
        out = torch.empty(a1.shape[:-1], dtype=torch.float32)
        _hip.check(_hip.lib().y2_iou_pair_host(a1.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr(), out.numel(), min, out.data_ptr()), 'y2_iou_pair_host')
        return out
    _hip.require_gpu(yx_min1, yx_max1, yx_min2, yx_max2)
    a1, b1, a2, b2 = (_hip.f32c(t) for t in (yx_min1, yx_max1, yx_min2, yx_max2))
    out = torch.empty(a1.shape[:-1], dtype=torch.float32, device=a1.device)
    _hip.check(_hip.lib().y2_iou_pair(_hip.ptr(a1), _hip.ptr(b1), _hip.ptr(a2), _hip.ptr(b2), out.numel(), min,
                                      _hip.ptr(out), _hip.stream()), 'y2_iou_pair')
    return out


def iou_rowmax(yx_min1, yx_max1, yx_min2, yx_max2, min=EPS):
    """(max over boxes 2 of the IoU, its first arg-max) per box 1 without the matrix in memory: `iou_matrix(...).max(-1)` of eval.py:69-70
    as one kernel (y2_iou_rowmax).  GPU tensors; N2 >= 1."""
    _hip.require_gpu(yx_min1, yx_max1, yx_min2, yx_max2)
    a1, b1, a2, b2 = (_hip.f32c(t) for t in (yx_min1, yx_max1, yx_min2, yx_max2))
    n1, n2 = a1.size(0), a2.size(0)
    best = torch.empty(n1, dtype=torch.float32, device=a1.device)
    which = torch.empty(n1, dtype=torch.int64, device=a1.device)
    _hip.check(_hip.lib().y2_iou_rowmax(_hip.ptr(a1), _hip.ptr(b1), _hip.ptr(a2), _hip.ptr(b2), n1, n2, min, _hip.ptr(best), _hip.ptr(which), _hip.stream()), 'y2_iou_rowmax')
    return best, which
