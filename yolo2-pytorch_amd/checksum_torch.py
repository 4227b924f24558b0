#!/usr/bin/env python
"""`checksum_torch` — abs-mean + md5 fingerprint of a checkpoint and of the backbone's output on a seeded random image.

Mirror of the reference's checksum_torch.py:44-65: one tab-separated row `key, shape, abs-mean, md5(raw bytes)` per
state_dict entry, then the same for the input `tensor` (torch.randn(1,3,H,W) under `torch.manual_seed(seed)`, :44,:55) and
the plugin's `output`.  Parameter rows are bit-comparable with the reference's printout for the same checkpoint (same
bytes); the `output` row comes from the MI355X kernels, so its abs-mean agrees to fp32 tolerance and its md5 differs
(different summation order) - `compare_rows` does exactly that comparison.
"""
import argparse
import configparser
import glob
import hashlib
import os

import numpy as np
import torch

from convert_darknet_torch import abs_mean


def row(key, a):
    a = np.ascontiguousarray(a)
    return '\t'.join(map(str, [key, a.shape, abs_mean(a), hashlib.md5(a.tobytes()).hexdigest()]))


def checksum_rows(dnn, tensor):
    """Rows for every state_dict entry of `dnn`, then `tensor` and `output = dnn(tensor)` (eval mode, no grad).  `tensor` is
    moved to the plugin's device; the plugin runs wherever its parameters live (GPU for the HIP-backed classes)."""
    rows = [row(key, var.detach().cpu().numpy()) for key, var in dnn.state_dict().items()]
    dev = next(dnn.parameters()).device
    dnn.eval()
    with torch.no_grad():
        output = dnn(tensor.to(dev))
    rows.append(row('tensor', tensor.cpu().numpy()))
    rows.append(row('output', output.cpu().numpy()))
    return rows


def compare_rows(ours, theirs, rtol=1e-4, skip=('num_batches_tracked',)):
    """[(key, what)] for every row of `theirs` (e.g. the reference's printout) that `ours` does not reproduce: parameter and
    input rows must match exactly (shape, abs-mean text, md5), the `output` row in shape and abs-mean within rtol."""
    mine = {r.split('\t')[0]: r.split('\t') for r in ours}
    bad = []
    for r in theirs:
        f = r.split('\t')
        key = f[0]
        if any(s in key for s in skip):
            continue
        if key not in mine:
            bad.append((key, 'missing'))
        elif key == 'output':
            if mine[key][1] != f[1] or abs(float(mine[key][2]) - float(f[2])) > rtol * abs(float(f[2])):
                bad.append((key, 'output differs: %s vs %s' % (mine[key][1:3], f[1:3])))
        elif mine[key] != f:
            bad.append((key, 'differs'))
    return bad


def main():
    import model
    import utils
    args = make_args()
    config = configparser.ConfigParser()
    utils.load_config(config, args.config)
    for cmd in args.modify:
        utils.modify_config(config, cmd)
    torch.manual_seed(args.seed)
    cache_dir = utils.get_cache_dir(config)
    model_dir = utils.get_model_dir(config)
    category = utils.get_category(config, cache_dir if os.path.exists(cache_dir) else None)
    anchors = torch.from_numpy(utils.get_anchors(config)).contiguous()
    # latest `<step>.pth` of the model directory (utils.train.load_model, utils/train.py:51-76)
    steps = [(int(os.path.splitext(os.path.basename(p))[0]), p) for p in glob.glob(os.path.join(model_dir, '*.pth')) if os.path.splitext(os.path.basename(p))[0].isdigit()]
    path = max(steps)[1]
    state_dict = torch.load(path, map_location='cpu')
    dnn = utils.parse_attr(config.get('model', 'dnn'))(model.ConfigChannels(config, state_dict), anchors, len(category))
    dnn.load_state_dict(state_dict, strict=False)       # torch >= 0.4 adds bn.num_batches_tracked, absent from converted checkpoints
    height, width = tuple(map(int, config.get('image', 'size').split()))
    tensor = torch.randn(1, 3, height, width)
    if torch.cuda.is_available():
        dnn.cuda()
    for r in checksum_rows(dnn, tensor):
        print(r)


def make_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', nargs='+', default=['config.ini'], help='config file')
    parser.add_argument('-m', '--modify', nargs='+', default=[], help='modify config')
    parser.add_argument('-s', '--seed', default=0, type=int, help='a seed to create a random image tensor')
    return parser.parse_args()


if __name__ == '__main__':
    main()
