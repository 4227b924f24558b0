"""`eval` — the IoU-matching step of the reference's VOC evaluation (eval.py:57-75), on the MI355X IoU kernel.

Only `matching` / `_matching` are on the hot path (another `iou_matrix` + arg-max consumer); the mAP
bookkeeping, TinyDB / xlsx reporting and dataset loop of eval.py are host-side harness code and out of scope."""
import numpy as np
import torch

import utils.iou.torch


def _matching(positive, index):
    """eval.py:57-64: greedy true-positive assignment, each ground-truth box is detected at most once."""
    detected = set()
    tp = np.zeros([len(positive)], bool)
    for i, (positive, index) in enumerate(zip(positive, index)):
        if positive and index not in detected:
            tp[i] = True
            detected.add(index)
    return tp


def matching(data_yx_min, data_yx_max, yx_min, yx_max, threshold):
    """eval.py:67-75: predictions (in descending-score order) vs ground truth of one class in one image."""
    if data_yx_min.numel() > 0:
        matrix = utils.iou.torch.iou_matrix(yx_min, yx_max, data_yx_min, data_yx_max)
        iou, index = torch.max(matrix, -1)
        positive = iou > threshold
        tp = _matching(positive.cpu().numpy(), index.cpu().numpy())
    else:
        tp = np.zeros([yx_min.size(0)], bool)
    return tp
