"""numpy restatement of the detection post-filter (detect.py:43-80).

filter_visible (detect.py:51-63) -> nms (utils/postprocess.py:23-49) ->
per-class score expansion (detect.py:72-79).  `prob` is softmax(logits)
computed by the caller (detect.py:152, eval.py:270).
"""
import numpy as np

from . import nms as _nms


def softmax(logits):
    """detect.py:152 `F.softmax(logits, -1)` — fp32, max-subtracted like ATen."""
    logits = np.asarray(logits, dtype=np.float32)
    m = logits.max(-1, keepdims=True)
    e = np.exp(logits - m, dtype=np.float32)
    return e / e.sum(-1, keepdims=True, dtype=np.float32)


def filter_visible(iou, yx_min, yx_max, prob, fix, threshold, threshold_cls):
    """detect.py:51-63.  iou [n], yx_* [n,2], prob [n,C]."""
    iou = np.asarray(iou, np.float32).reshape(-1)
    yx_min = np.asarray(yx_min, np.float32).reshape(-1, 2)
    yx_max = np.asarray(yx_max, np.float32).reshape(-1, 2)
    prob = np.asarray(prob, np.float32).reshape(iou.size, -1)
    cls = prob.argmax(-1)  # first maximal index, :52
    prob_cls = prob[np.arange(iou.size), cls]
    if fix:
        mask = (iou * prob_cls) > np.float32(threshold_cls)  # :54
    else:
        mask = iou > np.float32(threshold)  # :56
    idx = np.nonzero(mask)[0]  # boolean compaction keeps candidate order, :57-62
    return iou[idx], yx_min[idx], yx_max[idx], prob[idx], prob_cls[idx], cls[idx], idx


def postprocess(iou, yx_min, yx_max, prob, fix=False, threshold=0.3, threshold_cls=0.005, overlap=0.45, limit=200):
    """detect.py:66-80.  Returns None when nothing survives, else
    (iou, yx_min, yx_max, cls, score, keep_indices_into_the_unfiltered_input)."""
    iou, yx_min, yx_max, prob, prob_cls, cls, idx = filter_visible(iou, yx_min, yx_max, prob, fix, threshold, threshold_cls)
    keep = _nms.nms(iou, yx_min, yx_max, overlap, limit)  # :68
    if not keep:
        return None
    keep = np.asarray(keep, np.int64)
    iou, yx_min, yx_max, prob, prob_cls, cls = (t[keep] for t in (iou, yx_min, yx_max, prob, prob_cls, cls))
    src = idx[keep]
    if fix:
        score = iou[:, None] * prob  # :73
        mask = score > np.float32(threshold_cls)  # :74
        indices, cls = np.nonzero(mask)  # row-major (box-major, class-minor), :75
        yx_min, yx_max = yx_min[indices], yx_max[indices]
        score = score[mask]
        src = src[indices]
    else:
        score = iou  # :79
    return iou, yx_min, yx_max, cls.astype(np.int64), score, src
