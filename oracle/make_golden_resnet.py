"""Generate tests/golden/resnet.npz from the REFERENCE model/resnet.py (build container only).

    python -m oracle.make_golden_resnet

torchvision is not installed; the reference file only needs `torchvision.models.resnet.{ResNet, conv3x3, model_urls}`
for lineage (model/resnet.py:23-24,107), so a 6-line stub module is injected while the file is loaded by path.  The
script also asserts that oracle/resnet.py reproduces the reference output exactly (same torch kernels)."""
import configparser
import importlib.util
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refload, synth  # noqa: E402
from oracle import resnet as ores  # noqa: E402

CASES = (('resnet50', 8, 96, 80, 2), ('resnet18', 8, 64, 20, 2), ('resnet50', 64, 64, 80, 1))   # arch, width, size, classes, batch


def load_reference_resnet(ns):
    tv, tvm, tvr = types.ModuleType('torchvision'), types.ModuleType('torchvision.models'), types.ModuleType('torchvision.models.resnet')

    class _R(nn.Module):
        pass
    tvr.ResNet, tvr.model_urls = _R, {}
    tvr.conv3x3 = lambda i, o, stride=1: nn.Conv2d(i, o, 3, stride, 1, bias=False)
    tv.models, tvm.resnet = tvm, tvr
    names = {'torchvision': tv, 'torchvision.models': tvm, 'torchvision.models.resnet': tvr, 'model': ns.model}
    sys.modules.update(names)
    try:
        spec = importlib.util.spec_from_file_location('_ref_resnet', os.path.join(refload.REF, 'model/resnet.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k in names:
            sys.modules.pop(k, None)
    return m


def main():
    logging.disable(logging.WARNING)
    ns = refload.load()
    m = load_reference_resnet(ns)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(refload.REF, 'config.ini'))
    cfg.set('model', 'pretrained', '0')
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    out = {}
    for arch, width, S, C, B in CASES:
        sd = ores.init_state_dict(arch, 5, C, seed=0, width=width, head_scale=0.25)
        net = getattr(m, arch)(ns.model.ConfigChannels(cfg, sd), anchors, C)
        r = net.load_state_dict(sd, strict=False)
        assert not r.unexpected_keys and all(k.endswith('num_batches_tracked') for k in r.missing_keys), r
        net.eval()
        x = synth.images(B, S, seed=1)
        with torch.no_grad():
            f = net(x)
            assert torch.equal(f, ores.forward(x, sd, arch)), 'oracle/resnet.py deviates from the reference'
            f64 = ores.forward(x.double(), {k: v.double() for k, v in sd.items()}, arch)
        out['%s_w%d_feature' % (arch, width)] = f.numpy()
        out['%s_w%d_fp64' % (arch, width)] = f64.numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'resnet.npz'), **out)
    print('wrote tests/golden/resnet.npz')


if __name__ == '__main__':
    main()
