"""`utils.postprocess` — greedy NMS, MI355X-native (utils/postprocess.py:23-49).

`nms` keeps the reference signature and returns a Python list of indices (descending score); it accepts GPU tensors
(y2_nms) and CPU tensors (y2_nms_host: the reference's summary worker calls it on CPU tensors, train.py:209).  `nms_batch` is
the device-resident form used by detect.postprocess_batch: all images of a batch in one launch pair, no host
round trip.
"""
import torch

import _hip


def nms_batch(score, yx_min, yx_max, n, overlap=0.5, limit=200, cand=None):
    """score [B,stride], yx_min/yx_max [B,stride,2], n int32 [B] (valid candidates per image), optional
    cand int32 [B,stride] (candidate i of image b = row cand[b,i]; the output of detect.filter_visible_batch).
    Returns (keep int32 [B,limit] = positions in the candidate list, keep_count int32 [B]) on the GPU."""
    _hip.require_gpu(score, yx_min, yx_max, n)
    score, yx_min, yx_max = (_hip.f32c(t) for t in (score, yx_min, yx_max))
    B, stride = score.shape
    dev = score.device
    n = n.to(torch.int32).contiguous()
    order = torch.empty(B, limit, dtype=torch.int32, device=dev)
    keep = torch.empty(B, limit, dtype=torch.int32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    _hip.check(_hip.lib().y2_nms(_hip.ptr(score), _hip.ptr(yx_min), _hip.ptr(yx_max), _hip.ptr(cand), _hip.ptr(n), B, stride, overlap, limit,
                                 _hip.ptr(order), _hip.ptr(keep), _hip.ptr(cnt), _hip.stream()), 'y2_nms')
    return keep, cnt


def _nms_host(score, yx_min, yx_max, overlap, limit):
    """CPU tensors (the reference's summary worker, train.py:209): y2_nms_host, the library's host implementation of the same
    rank + greedy algorithm as the device kernels (bit-identical keep lists; no HIP call, safe in a forked child)."""
    score, yx_min, yx_max = (t.detach().to(torch.float32).contiguous() for t in (score.reshape(-1), yx_min.reshape(-1, 2), yx_max.reshape(-1, 2)))
    n = torch.tensor([score.numel()], dtype=torch.int32)
    keep = torch.empty(limit, dtype=torch.int32)
    cnt = torch.empty(1, dtype=torch.int32)
    _hip.check(_hip.lib().y2_nms_host(score.data_ptr(), yx_min.data_ptr(), yx_max.data_ptr(), None, n.data_ptr(), 1, score.numel(), overlap, limit,
                                      keep.data_ptr(), cnt.data_ptr()), 'y2_nms_host')
    return keep[:int(cnt[0])].tolist()


def nms(score, yx_min, yx_max, overlap=0.5, limit=200):
    """utils/postprocess.py:23-49: indices of the selected boxes, in descending-score order.  GPU tensors run y2_nms, CPU
    tensors y2_nms_host (same library, same algorithm)."""
    keep = []
    if score.numel() == 0:
        return keep
    if not score.is_cuda:
        return _nms_host(score, yx_min, yx_max, overlap, limit)
    n = torch.tensor([score.numel()], dtype=torch.int32, device=score.device)
    k, c = nms_batch(score.reshape(1, -1), yx_min.reshape(1, -1, 2), yx_max.reshape(1, -1, 2), n, overlap, limit)
    return k[0, :int(c.item())].tolist()
