#!/usr/bin/env python
"""`convert_darknet_torch` — import an original Darknet `.weights` file into the HIP-backed plugin's state_dict.

Mirror of the reference's convert_darknet_torch.py (the only real-weights parity route).  The plugin
classes of this tree keep the reference's state_dict keys, shapes and ORDER, so the file walks the same way:

  * header: four little-endian int32 `major, minor, revision, seen` (convert_darknet_torch.py:92);
  * then fp32 values, layer by layer in state_dict order (layer = key minus its last two components, :62-71), inside a
    layer in Darknet's order `conv.bias | bn.bias, bn.weight, bn.running_mean, bn.running_var`, then `conv.weight`
    [Cout,Cin,k,k] (:99);
  * Darknet's region layer lays each anchor's rows out as (x, y, w, h, objectness, classes...) while model.Inference decodes
    (iou, y, x, h, w, classes...) (model/__init__.py:123-135): the rows of the LAST layer's weight and bias are permuted
    accordingly (:37-59, :113-114).

`load_darknet_weights` is the library function (numpy `frombuffer`, no per-float unpacking); `main` keeps the reference's
command line (`file -c -m -d --copy`) and writes `<model_dir>/0.pth` + `0.epoch` like utils.train.Saver (utils/train.py:95-111).
Host-side code: nothing here touches the GPU.
"""
import argparse
import collections
import configparser
import hashlib
import logging
import os
import shutil
import struct

import numpy as np
import torch

FILE_SUFFIXES = ('conv.bias', 'bn.bias', 'bn.weight', 'bn.running_mean', 'bn.running_var', 'conv.weight')
_DARKNET_TO_MODEL = (4, 1, 0, 3, 2)      # model row j of an anchor <- darknet row: iou<-objectness, y<-y, x<-x, h<-h, w<-w


def _head_permutation(rows, num_anchors):
    per = rows // num_anchors
    if per * num_anchors != rows or per < 5:
        raise ValueError('head has %d rows: not %d anchors x (5 + classes)' % (rows, num_anchors))
    one = list(_DARKNET_TO_MODEL) + list(range(5, per))
    return torch.tensor([a * per + j for a in range(num_anchors) for j in one], dtype=torch.long)


def transpose_weight(weight, num_anchors):
    """Region-layer weight rows, Darknet order -> model order (convert_darknet_torch.py:37-46)."""
    return weight.index_select(0, _head_permutation(weight.size(0), num_anchors))


def transpose_bias(bias, num_anchors):
    """Region-layer bias, Darknet order -> model order (convert_darknet_torch.py:49-57)."""
    return bias.index_select(0, _head_permutation(bias.size(0), num_anchors))


def group_state(state_dict):
    """layer -> {suffix: tensor}, layers in first-appearance order (convert_darknet_torch.py:60-69)."""
    grouped = collections.OrderedDict()
    for key, var in state_dict.items():
        layer, s1, s2 = key.rsplit('.', 2)
        grouped.setdefault(layer, {})[s1 + '.' + s2] = var
    return grouped


def abs_mean(a):
    """utils.abs_mean (utils/__init__.py:119-121)."""
    return np.sum(np.abs(a)) / np.float32(a.size)


def load_darknet_weights(path, state_dict, num_anchors, log=None):
    """Read `path` into an OrderedDict with the reference converter's keys and ORDER (file order, head rows permuted).
    `state_dict` supplies key order and shapes (the plugin's `state_dict()`).  Returns (converted, info) with
    info = dict(major, minor, revision, seen, assigned, remaining)."""
    with open(os.path.expanduser(os.path.expandvars(path)), 'rb') as f:
        data = f.read()
    if len(data) < 16:
        raise ValueError('%s: shorter than the 16-byte Darknet header' % path)
    major, minor, revision, seen = struct.unpack('<4i', data[:16])
    pos, total = 16, 0
    layers = []
    for layer, group in group_state(state_dict).items():
        for suffix in FILE_SUFFIXES:
            if suffix not in group:
                continue
            shape = tuple(group[suffix].shape)
            cnt = int(np.prod(shape, dtype=np.int64))
            if pos + 4 * cnt > len(data):
                raise ValueError('%s ends inside %s.%s (%d floats wanted, %d bytes left)' % (path, layer, suffix, cnt, len(data) - pos))
            val = np.frombuffer(data, dtype='<f4', count=cnt, offset=pos).reshape(shape).copy()
            pos += 4 * cnt
            total += cnt
            if log is not None:
                log('%s.%s: %s=%f (%s), remaining=%d' % (layer, suffix, 'x'.join(map(str, shape)), abs_mean(val), hashlib.md5(val.tobytes()).hexdigest(), len(data) - pos))
            layers.append([layer + '.' + suffix, torch.from_numpy(val)])
    if len(layers) < 2:
        raise ValueError('state_dict has no convolution layers')
    layers[-1][1] = transpose_weight(layers[-1][1], num_anchors)
    layers[-2][1] = transpose_bias(layers[-2][1], num_anchors)
    info = dict(major=major, minor=minor, revision=revision, seen=seen, assigned=total, remaining=len(data) - pos)
    return collections.OrderedDict(layers), info


def save_checkpoint(state_dict, model_dir, step=0, epoch=0):
    """`<model_dir>/<step>.pth` + `.epoch`, the layout utils.train.load_model reads back (utils/train.py:51-76, 95-111)."""
    os.makedirs(model_dir, exist_ok=True)
    prefix = os.path.join(model_dir, str(step))
    torch.save(state_dict, prefix + '.pth')
    with open(prefix + '.epoch', 'w') as f:
        f.write(str(epoch))
    return prefix + '.pth'


def main():
    import model
    import utils
    args = make_args()
    config = configparser.ConfigParser()
    utils.load_config(config, args.config)
    for cmd in args.modify:
        utils.modify_config(config, cmd)
    logging.basicConfig(level=logging.INFO)
    cache_dir = utils.get_cache_dir(config)
    model_dir = utils.get_model_dir(config)
    category = utils.get_category(config, cache_dir if os.path.exists(cache_dir) else None)
    anchors = torch.from_numpy(utils.get_anchors(config)).contiguous()
    dnn = utils.parse_attr(config.get('model', 'dnn'))(model.ConfigChannels(config), anchors, len(category))
    converted, info = load_darknet_weights(args.file, dnn.state_dict(), len(anchors), log=logging.info)
    logging.info('major=%(major)d, minor=%(minor)d, revision=%(revision)d, seen=%(seen)d; %(assigned)d parameters assigned' % info)
    if info['remaining'] > 0:
        logging.warning('%d bytes remaining' % info['remaining'])
    if args.delete:
        logging.warning('delete model directory: ' + model_dir)
        shutil.rmtree(model_dir, ignore_errors=True)
    path = save_checkpoint(converted, model_dir)
    if args.copy is not None:
        dst = os.path.expandvars(os.path.expanduser(args.copy))
        logging.info('copy %s to %s' % (path, dst))
        shutil.copy(path, dst)


def make_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('file', help='Darknet .weights file')
    parser.add_argument('-c', '--config', nargs='+', default=['config.ini'], help='config file')
    parser.add_argument('-m', '--modify', nargs='+', default=[], help='modify config')
    parser.add_argument('-d', '--delete', action='store_true', help='delete logdir')
    parser.add_argument('--copy', help='copy model')
    return parser.parse_args()


if __name__ == '__main__':
    main()
