"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/yolo2_hip.h declares; the product refuses CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

import _hip


@pytest.fixture(scope='module')
def built():
    if not os.path.exists(_hip.LIB_PATH):
        _hip.build()
    return _hip.LIB_PATH


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'yolo2_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(y2_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(built):
    l = ctypes.CDLL(built)
    names = declared_symbols()
    assert len(names) >= 13
    for name in names:
        assert hasattr(l, name), name
    # and the Python binding knows every compute entry point
    for name in names:
        assert name in _hip.SIGNATURES or name == 'y2_build_info', name


def test_version_and_build_info(built):
    l = _hip.lib()
    assert l.y2_abi_version() == 2
    assert b'gfx950' in l.y2_build_info()


def test_conv_params_struct_layout():
    # 7 pointers + 12 int32 + 1 float + 1 int32 (112 bytes, 8-aligned) + workspace pointer + int64 size + residual pointer + 7 int32 (+ pad) + int64 w_plane
    assert ctypes.sizeof(_hip.ConvParams) == 176


def test_no_cpu_fallback_on_the_gpu_path():
    """The convolution / decode / loss path refuses CPU tensors (only nms and the IoU helpers have host entry points, because
    the reference calls those on CPU tensors: train.py:209, utils/iou/torch.py:64-113)."""
    import configparser

    import detect
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.ones(5, 2)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg), anchors, 20).eval()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dnn(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model.decode(torch.zeros(1, 1, 1, 125), anchors, 5)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        detect.filter_visible_batch(torch.zeros(1, 5), torch.zeros(1, 5), True, 0.005)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(_hip, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.lib()
