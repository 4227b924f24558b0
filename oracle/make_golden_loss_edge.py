"""Edge-case region-loss fixture generated from the REFERENCE (patched as in oracle/make_golden.py).

    python -m oracle.make_golden_loss_edge          # needs /root/reference; writes tests/golden/loss_edge.npz

Cases the reference's data contract (utils/data.py:29-42: zero-padded boxes) allows and the ordinary fixture does not
hold: an image with no object at all, two identical boxes, two different boxes whose centres fall into one cell
(collision on the cell, possibly on the anchor), a box whose centre lies in the last row/column, a box clipped at the
image border, a degenerate (zero-height) box at the head of the list; on the 10x10, 13x13 and 19x19 grids of the
reference's multi-scale training sizes (config.ini [data] sizes are square; on a non-square grid the reference itself
returns an infinite size loss for these boxes, so that case is not a parity target).
Inputs are rebuilt from oracle/synth.py `edge_labels` / `edge_feature`, so the fixture holds OUTPUTS only.
"""
import os

import numpy as np
import torch

from oracle import make_golden as mg
from oracle import refload, synth

WEIGHTS = dict(foreground=5, background=1, center=1, size=1, cls=1)


CASES = (('sq13', 416, 416, 13, 13), ('sq19', 608, 608, 19, 19), ('sq10', 320, 320, 10, 10))


def main():
    assert refload.available(), 'reference not mounted'
    ns = refload.load()
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    mp = mg.patched_loss_module(ns)
    res = {}
    for name, S_h, S_w, rows, cols in CASES:
        for onehot in (False, True):
            feat = synth.edge_feature(5, 5, 20, rows, cols).requires_grad_(True)

            class Id(torch.nn.Module):
                def forward(self, t):
                    return t
            inf = mp.Inference(mg.ref_config(), Id(), anchors)
            pred = mp._inference(inf, feat)
            data = synth.norm_data(synth.edge_labels(S_h, S_w, 20, onehot), S_h, S_w, rows, cols)
            loss, debug = mp.loss(anchors, data, pred, 0.6)
            tot = sum(loss[k] * w for k, w in WEIGHTS.items())
            tot.backward()
            tag = '%s_%s_' % (name, 'onehot' if onehot else 'ce')
            for k, v in loss.items():
                res[tag + k] = v.detach().numpy()
            res[tag + 'grad'] = feat.grad.numpy()
            res[tag + 'positive'] = debug['positive'].numpy()
            res[tag + 'negative'] = debug['negative'].numpy()
            res[tag + 'best_iou'] = debug['iou'].numpy()
            print(tag, {k: float(v) for k, v in loss.items()}, 'positives', int(debug['positive'].sum()),
                  'finite grad', bool(torch.isfinite(feat.grad).all()))
    np.savez_compressed(os.path.join(mg.OUT, 'loss_edge.npz'), **res)


if __name__ == '__main__':
    main()
