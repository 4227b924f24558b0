"""numpy restatement of the reference's greedy NMS (bit-exact index spec).

Follows utils/postprocess.py:23-49: sort scores descending, keep the first
`limit`, then repeatedly take the head and drop every remaining box whose IoU
with it is > overlap (kept iff `iou <= overlap`, compared in fp32).

Tie rule: the reference's `sort(descending=True)` is not stable, so equal
scores have implementation-defined order there.  This
oracle — and the HIP kernel — define ties as lower-original-index-first.
"""
import numpy as np

from . import iou as _iou


def nms(score, yx_min, yx_max, overlap=0.5, limit=200):
    score = np.ascontiguousarray(score, dtype=np.float32).reshape(-1)
    yx_min = np.ascontiguousarray(yx_min, dtype=np.float32).reshape(-1, 2)
    yx_max = np.ascontiguousarray(yx_max, dtype=np.float32).reshape(-1, 2)
    keep = []
    if score.size == 0:  # :35-36
        return keep
    index = np.argsort(-score, kind='stable')[:limit]  # :37-38
    thr = np.float32(overlap)
    while index.size > 0:  # :39
        i = int(index[0])
        keep.append(i)  # :41
        if index.size == 1:  # :42-43
            break
        index = index[1:]
        iou = _iou.iou_matrix(yx_min[i:i + 1], yx_max[i:i + 1], yx_min[index], yx_max[index])[0]  # :45-47
        index = index[iou <= thr]  # :48
    return keep
