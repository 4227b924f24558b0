"""`utils` — config / plugin-resolution helpers with the reference's names (utils/__init__.py:53-112)."""
import configparser
import importlib
import os

import numpy as np
import torch


def get_cache_dir(config):
    root = os.path.expanduser(os.path.expandvars(config.get('config', 'root')))
    return os.path.join(root, config.get('cache', 'name'))


def get_model_dir(config):
    root = os.path.expanduser(os.path.expandvars(config.get('config', 'root')))
    return os.path.join(root, config.get('model', 'name'), config.get('model', 'dnn'))


def get_category(config, cache_dir=None):
    path = os.path.expanduser(os.path.expandvars(config.get('cache', 'category'))) if cache_dir is None else os.path.join(cache_dir, 'category')
    with open(path, 'r') as f:
        return [line.strip() for line in f]


def get_anchors(config, dtype=np.float32):
    """utils/__init__.py:78-81: the tsv has columns width,height; anchors are returned as (height, width)."""
    path = os.path.expanduser(os.path.expandvars(config.get('model', 'anchors')))
    with open(path) as f:
        header = f.readline().strip().split('\t')
        rows = [[float(v) for v in line.strip().split('\t')] for line in f if line.strip()]
    a = np.array(rows, dtype=dtype)
    return a[:, [header.index('height'), header.index('width')]]


def parse_attr(s):
    """utils/__init__.py:84-87: 'pkg.mod.Name' -> attribute (the plugin mechanism of config.ini)."""
    m, n = s.rsplit('.', 1)
    return getattr(importlib.import_module(m), n)


def load_config(config, paths):
    for path in paths:
        path = os.path.expanduser(os.path.expandvars(path))
        assert os.path.exists(path), path
        config.read(path)


def modify_config(config, cmd):
    var, value = cmd.split('=', 1)
    section, option = var.split('/')
    if value:
        config.set(section, option, value)
    else:
        try:
            config.remove_option(section, option)
        except (configparser.NoSectionError, configparser.NoOptionError):
            pass


def ensure_device(t, device_id=None, non_blocking=False):
    """utils/__init__.py:109-112 (`async` is a keyword since Python 3.7)."""
    if torch.cuda.is_available():
        t = t.cuda(device_id, non_blocking)
    return t
