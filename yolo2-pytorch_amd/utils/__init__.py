"""`utils` — config and plugin-resolution helpers under the reference's names (utils/__init__.py:53-112).

Same call signatures and ini keys as the reference, so an unmodified `config.ini` resolves `[model] dnn =
model.yolo2.Darknet` (or `model.resnet.resnet50`, `model.yolo2.Tiny`) to the HIP-backed plugin classes of this tree.
"""
import configparser
import importlib
import os

import numpy as np
import torch


def _path(p):
    return os.path.expanduser(os.path.expandvars(p))


def _root(config):
    return _path(config.get('config', 'root'))


def get_cache_dir(config):
    """<root>/<[cache] name>  (utils/__init__.py:53-56)."""
    return os.path.join(_root(config), config.get('cache', 'name'))


def get_model_dir(config):
    """<root>/<[model] name>/<[model] dnn>  (utils/__init__.py:59-63)."""
    return os.path.join(_root(config), config.get('model', 'name'), config.get('model', 'dnn'))


def get_eval_db(config):
    return os.path.join(_root(config), config.get('eval', 'db'))


def get_category(config, cache_dir=None):
    """Class names, one per line (utils/__init__.py:72-75)."""
    path = _path(config.get('cache', 'category')) if cache_dir is None else os.path.join(cache_dir, 'category')
    with open(path) as f:
        return [line.strip() for line in f]


def get_anchors(config, dtype=np.float32):
    """Anchor table in cell units as (height, width) rows; the tsv columns are `width  height` (utils/__init__.py:78-81)."""
    with open(_path(config.get('model', 'anchors'))) as f:
        names = f.readline().split()
        table = np.array([[float(v) for v in line.split()] for line in f if line.strip()], dtype=dtype)
    return table[:, [names.index('height'), names.index('width')]]


def parse_attr(s):
    """'package.module.Name' -> the attribute: the plugin mechanism of config.ini (utils/__init__.py:84-87)."""
    module, _, name = s.rpartition('.')
    return getattr(importlib.import_module(module), name)


def load_config(config, paths):
    """Overlay ini files in order (utils/__init__.py:90-94)."""
    for p in map(_path, paths):
        if not os.path.exists(p):
            raise AssertionError(p)
        config.read(p)


def modify_config(config, cmd):
    """`section/option=value` sets, `section/option=` deletes (utils/__init__.py:97-106)."""
    target, value = cmd.split('=', 1)
    section, option = target.split('/')
    if value:
        config.set(section, option, value)
        return
    try:
        config.remove_option(section, option)
    except (configparser.NoSectionError, configparser.NoOptionError):
        pass


def ensure_device(t, device_id=None, non_blocking=False):
    """utils/__init__.py:109-112 (`async` became a keyword in Python 3.7; the argument is now `non_blocking`)."""
    return t.cuda(device_id, non_blocking) if torch.cuda.is_available() else t


from . import optim  # noqa: E402,F401  (utils.optim.SGD / Adam / clip_grad_norm_ for the ini lambda, config.ini:72)
