"""numpy restatement of the reference's Darknet `.weights` importer and checksum printout — TEST INFRASTRUCTURE.

File format as read by convert_darknet_torch.py:92-113: a 16-byte header of four little-endian int32 (major, minor,
revision, seen), then fp32 values consumed in `state_dict()` key order grouped per layer (`key.rsplit('.', 2)`,
:62-71), and inside each layer in the order conv.bias | bn.bias, bn.weight, bn.running_mean, bn.running_var, conv.weight
(:99).  Darknet stores the region layer's rows per anchor as (x, y, w, h, objectness, classes...); the reference wants
(iou, y, x, h, w, classes...) (model/__init__.py:123-135), hence the permutation of the LAST layer's weight and bias rows
(:37-59, applied at :113-114).  Checksum rows follow checksum_torch.py:55-65 with utils.abs_mean (utils/__init__.py:119-121).

Pinned by tests/golden/darknet_weights.npz, produced by running the reference's own main() functions
(oracle/make_golden_darknet_weights.py).
"""
import collections
import hashlib
import struct

import numpy as np

SUFFIXES = ['conv.bias', 'bn.bias', 'bn.weight', 'bn.running_mean', 'bn.running_var', 'conv.weight']  # convert_darknet_torch.py:99


def group_keys(keys):
    """convert_darknet_torch.py:62-71 on key names: OrderedDict layer -> [suffix, ...] in first-appearance order."""
    grouped = collections.OrderedDict()
    for key in keys:
        layer, s1, s2 = key.rsplit('.', 2)
        grouped.setdefault(layer, []).append(s1 + '.' + s2)
    return grouped


def file_order(shapes):
    """[(key, shape)] in the order the importer consumes the file; `shapes` = OrderedDict key -> shape (state_dict order)."""
    out = []
    for layer, suffixes in group_keys(shapes.keys()).items():
        for suffix in SUFFIXES:
            if suffix in suffixes:
                out.append((layer + '.' + suffix, tuple(shapes[layer + '.' + suffix])))
    return out


def permute_head_rows(a, num_anchors):
    """convert_darknet_torch.py:37-59 for a weight [A*(5+C), Cin, k, k] or a bias [A*(5+C)]: (x,y,w,h,iou,cls) -> (iou,y,x,h,w,cls)."""
    a = np.asarray(a)
    rest = a.shape[1:]
    v = a.reshape((num_anchors, -1) + rest)
    order = [4, 1, 0, 3, 2] + list(range(5, v.shape[1]))
    return np.ascontiguousarray(v[:, order]).reshape((-1,) + rest)


def read_weights(data, shapes, num_anchors):
    """bytes of a .weights file -> (OrderedDict key -> float32 array in FILE order, header tuple, remaining bytes)."""
    header = struct.unpack('<4i', data[:16])
    pos = 16
    out = collections.OrderedDict()
    for key, shape in file_order(shapes):
        cnt = int(np.prod(shape, dtype=np.int64))
        out[key] = np.frombuffer(data, '<f4', cnt, pos).reshape(shape).copy()
        pos += 4 * cnt
    keys = list(out.keys())
    out[keys[-1]] = permute_head_rows(out[keys[-1]], num_anchors)      # :113
    out[keys[-2]] = permute_head_rows(out[keys[-2]], num_anchors)      # :114
    return out, header, len(data) - pos


def write_weights(arrays_in_file_order, header=(0, 1, 0, 0)):
    """Inverse of the sequential read (no head permutation): header + concatenated fp32 values -> bytes."""
    return struct.pack('<4i', *header) + b''.join(np.ascontiguousarray(a, '<f4').tobytes() for a in arrays_in_file_order)


def abs_mean(a):
    """utils/__init__.py:119-121."""
    return np.sum(np.abs(a)) / np.float32(a.size)


def checksum_row(key, a):
    """checksum_torch.py:57 / :64: tab-separated key, shape, abs-mean, md5 of the raw bytes."""
    a = np.asarray(a)
    return '\t'.join(map(str, [key, a.shape, abs_mean(a), hashlib.md5(a.tobytes()).hexdigest()]))
