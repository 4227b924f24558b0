"""numpy fp32 restatement of the reference IoU functions (bit-exact spec).

Follows utils/iou/torch.py:24-61 (intersection_area / iou_matrix), :116-153
(batch variants) and :216-233 (batch_iou_pair).  All arithmetic is IEEE fp32,
one rounding per operation, no FMA contraction — the GPU kernels are compiled
with -ffp-contract=off to match this bit for bit.
"""
import numpy as np

EPS32 = np.float32(np.finfo(np.float32).eps)  # utils/iou/torch.py:47 default `min`


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def intersection_area(yx_min1, yx_max1, yx_min2, yx_max2):
    """utils/iou/torch.py:24-44 — [N1,2],[N1,2],[N2,2],[N2,2] -> [N1,N2]."""
    yx_min1, yx_max1, yx_min2, yx_max2 = map(_f32, (yx_min1, yx_max1, yx_min2, yx_max2))
    max_min = np.maximum(yx_min1[..., :, None, :], yx_min2[..., None, :, :])
    min_max = np.minimum(yx_max1[..., :, None, :], yx_max2[..., None, :, :])
    size = np.maximum(min_max - max_min, np.float32(0))  # clamp(min=0), :39,:42
    return size[..., 0] * size[..., 1]  # height * width, :43


def iou_matrix(yx_min1, yx_max1, yx_min2, yx_max2, min=EPS32):
    """utils/iou/torch.py:47-61."""
    yx_min1, yx_max1, yx_min2, yx_max2 = map(_f32, (yx_min1, yx_max1, yx_min2, yx_max2))
    inter = intersection_area(yx_min1, yx_max1, yx_min2, yx_max2)
    d1 = yx_max1 - yx_min1
    d2 = yx_max2 - yx_min2
    area1 = (d1[..., 0] * d1[..., 1])[..., :, None]
    area2 = (d2[..., 0] * d2[..., 1])[..., None, :]
    union = np.maximum((area1 + area2) - inter, np.float32(min))  # :60 (a1+a2) first, then -inter
    return inter / union


# the batched functions are the same math with a leading batch axis (:116-153)
batch_intersection_area = intersection_area
batch_iou_matrix = iou_matrix


def batch_iou_pair(yx_min1, yx_max1, yx_min2, yx_max2, min=EPS32):
    """utils/iou/torch.py:216-233 — elementwise IoU of paired boxes [N,M,2] -> [N,M]."""
    yx_min1, yx_max1, yx_min2, yx_max2 = map(_f32, (yx_min1, yx_max1, yx_min2, yx_max2))
    yx_min = np.maximum(yx_min1, yx_min2)
    yx_max = np.minimum(yx_max1, yx_max2)
    size = np.maximum(yx_max - yx_min, np.float32(0))
    inter = size[..., 0] * size[..., 1]
    d1 = yx_max1 - yx_min1
    d2 = yx_max2 - yx_min2
    union = np.maximum((d1[..., 0] * d1[..., 1] + d2[..., 0] * d2[..., 1]) - inter, np.float32(min))
    return inter / union
