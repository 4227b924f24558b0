"""`eval` — the IoU-matching step of the reference's VOC evaluation (eval.py:57-75), on the MI355X IoU kernel.

Only `matching` / `_matching` are on the hot path (another `iou_matrix` + arg-max consumer); the mAP
bookkeeping, TinyDB / xlsx reporting and dataset loop of eval.py are host-side harness code and out of scope."""
import numpy as np
import torch

import utils.iou.torch


def _matching(positive, index):
    """eval.py:57-64: walking the predictions in descending-score order, a prediction is a true positive when it overlaps a
    ground-truth box enough (`positive`) and that box (`index`) has not been claimed by an earlier prediction."""
    positive, index = np.asarray(positive, bool), np.asarray(index)
    tp = np.zeros(len(positive), bool)
    claimed = set()
    for i in np.flatnonzero(positive):
        gt = int(index[i])
        if gt not in claimed:
            claimed.add(gt)
            tp[i] = True
    return tp


def matching(data_yx_min, data_yx_max, yx_min, yx_max, threshold):
    """eval.py:67-75: predictions of one class in one image (descending score) against that class's ground truth.  The IoU matrix and
    its row arg-max run on the device in one kernel (y2_iou_rowmax); only two n-vectors cross to the host for the sequential claim loop."""
    if data_yx_min.numel() == 0:
        return np.zeros([yx_min.size(0)], bool)
    if not yx_min.is_cuda:
        best, which = utils.iou.torch.iou_matrix(yx_min, yx_max, data_yx_min, data_yx_max).max(-1)      # CPU tensors: the library's host IoU
    else:
        best, which = utils.iou.torch.iou_rowmax(yx_min, yx_max, data_yx_min, data_yx_max)               # one kernel: IoU row + max + first arg-max
    return _matching((best.cpu().numpy() > np.float32(threshold)), which.cpu().numpy())
