"""Generate tests/golden/tiny.npz from the REFERENCE model.yolo2.Tiny (build container only): python -m oracle.make_golden_tiny"""
import configparser
import logging
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import darknet as od  # noqa: E402
from oracle import refload, synth  # noqa: E402


def main():
    logging.disable(logging.WARNING)
    ns = refload.load()
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(refload.REF, 'config.ini'))
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    out = {}
    for div, S, B in ((8, 96, 2), (1, 64, 1)):
        sd = od.init_tiny_state_dict(5, 20, seed=0, div=div, head_scale=0.25)
        net = ns.yolo2.Tiny(ns.model.ConfigChannels(cfg, sd), anchors, 20)
        r = net.load_state_dict(sd, strict=False)
        assert not r.unexpected_keys and all(k.endswith('num_batches_tracked') for k in r.missing_keys), r
        net.eval()
        x = synth.images(B, S, seed=1)
        with torch.no_grad():
            f = net(x)
            assert torch.equal(f, od.tiny_forward(x, sd)), 'oracle tiny_forward deviates from the reference'
            f64 = od.tiny_forward(x.double(), {k: v.double() for k, v in sd.items()})
        out['tiny_div%d_feature' % div] = f.numpy()
        out['tiny_div%d_fp64' % div] = f64.numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'tiny.npz'), **out)


if __name__ == '__main__':
    main()
