"""torch-CPU restatement of the detection head decode (model/__init__.py:53-56, 110-135, 170-179)."""
import torch


def meshgrid(rows, cols):
    """model/__init__.py:53-56 (swap=False): entry k = (k // rows, k % rows) — equals
    (row, col) of cell k only when rows == cols (all shipped sizes are square)."""
    i = torch.arange(0, rows).repeat(cols).view(-1, 1)
    j = torch.arange(0, cols).view(-1, 1).repeat(1, rows).view(-1, 1)
    return torch.cat([j, i], 1)


def decode(feature, anchors):
    """model/__init__.py:117-135 Inference.forward after `self.dnn(x)`.
    feature [B,A(5+C),rows,cols], anchors [A,2] (h,w) -> dict like model._inference (:170-179)."""
    rows, cols = feature.shape[-2:]
    cells = rows * cols
    A = anchors.shape[0]
    _f = feature.permute(0, 2, 3, 1).contiguous().view(feature.shape[0], cells, A, -1)
    sig = torch.sigmoid(_f[:, :, :, :3])
    iou = sig[:, :, :, 0]
    ij = meshgrid(rows, cols).view(1, -1, 1, 2).to(feature.dtype)
    center_offset = sig[:, :, :, 1:3]
    center = ij + center_offset
    size_norm = _f[:, :, :, 3:5]
    size = torch.exp(size_norm) * anchors.view(1, 1, -1, 2).to(feature.dtype)
    size2 = size / 2
    pred = dict(feature=feature, iou=iou, center_offset=center_offset, size_norm=size_norm,
                yx_min=center - size2, yx_max=center + size2)
    if _f.shape[-1] > 5:
        pred['logits'] = _f[:, :, :, 5:].contiguous()
    return pred
