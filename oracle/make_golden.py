"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

    python -m oracle.make_golden            # needs /root/reference

The reference files are loaded by path (oracle/refload.py).  Two of them need
in-memory patching to run on torch 2.10 (never written back, never copied
into this repo):
  * model/__init__.py loss/fit_positive: six line-level edits restoring the
    torch-0.3 mask semantics.
  * detect.py: only lines 43-80 (get_logits / filter_visible / postprocess)
    are exec'd, with a no-op `pybenchmark.profile` (cv2 etc. are not installed).
Inputs are regenerated from seeds by oracle/synth.py + oracle/darknet.init_state_dict,
so the fixtures hold OUTPUTS only (small).
"""
import configparser
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import darknet as odark  # noqa: E402
from oracle import refload, synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

# narrow ("pruned-checkpoint") widths used for the small forward fixture
NARROW = {p: max(4, c // 16) for p, c in
          [(i[0], i[2]) for i in odark.LAYERS1 + odark.LAYERS2 + [odark.PASSTHROUGH] + odark.LAYERS3 if i != 'M']}
NARROW['layers1.5'] = 6  # deliberately not a multiple of 4: exercises the unaligned-channel path


def ref_config():
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(refload.REF, 'config.ini'))
    return cfg


def build_ref_model(ns, sd, num_cls, anchors):
    cfg = ref_config()
    cc = ns.model.ConfigChannels(cfg, sd)
    dnn = ns.yolo2.Darknet(cc, anchors, num_cls)
    missing = dnn.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    assert all(k.endswith('num_batches_tracked') for k in missing.missing_keys), missing
    inf = ns.model.Inference(cfg, dnn, anchors)
    return dnn, inf


def patched_loss_module(ns):
    src = open(os.path.join(refload.REF, 'model/__init__.py')).read()
    edits = [
        ('valid = torch.prod(yx_min < yx_max, -1)', 'valid = (yx_min < yx_max).all(-1)'),
        ('t = utils.ensure_device(torch.ByteTensor(cells, num_anchors).zero_(), device_id)', 't = torch.zeros(cells, num_anchors, dtype=torch.bool)'),
        ("pred['center_offset'][_positive], _center_offset[_positive]", "pred['center_offset'][_positive.expand_as(_center_offset)], _center_offset[_positive.expand_as(_center_offset)]"),
        ("pred['size_norm'][_positive], _size_norm[_positive]", "pred['size_norm'][_positive.expand_as(_size_norm)], _size_norm[_positive.expand_as(_size_norm)]"),
        ('F.softmax(logits, -1)[_positive], _cls[_positive]', 'F.softmax(logits, -1)[_positive.expand_as(logits)], _cls[_positive.expand_as(logits)]'),
        ('logits[_positive].view(-1, logits.size(-1))', 'logits[_positive.expand_as(logits)].view(-1, logits.size(-1))'),
    ]
    for a, b in edits:
        assert src.count(a) == 1, a
        src = src.replace(a, b)
    mod = types.ModuleType('_ref_model_patched')
    sys.modules['utils'] = ns.utils
    sys.modules['utils.iou'] = ns.utils.iou
    sys.modules['utils.iou.torch'] = ns.iou
    try:
        exec(compile(src, 'reference:model/__init__.py(patched)', 'exec'), mod.__dict__)
    finally:
        for k in ('utils', 'utils.iou', 'utils.iou.torch'):
            sys.modules.pop(k, None)
    return mod


def detect_functions(ns):
    lines = open(os.path.join(refload.REF, 'detect.py')).read().split('\n')[42:80]
    src = '\n'.join(lines)
    g = dict(torch=torch, utils=ns.utils,
             pybenchmark=types.SimpleNamespace(profile=lambda name: (lambda fn: fn)))
    exec(compile(src, 'reference:detect.py:43-80', 'exec'), g)
    return g


def det_config(fix):
    cfg = ref_config()
    cfg.set('detect', 'fix', str(int(fix)))
    return cfg


def main():
    assert refload.available(), 'reference not mounted'
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ns = refload.load()
    anchors = torch.from_numpy(synth.ANCHORS_VOC)

    # ---- 1. Darknet forward + decode, narrow widths (pruned-checkpoint path), S=96 (3x3 grid), B=2
    sd = odark.init_state_dict(5, 20, seed=0, channels=NARROW, head_scale=1 / 8.0)
    dnn, inf = build_ref_model(ns, sd, 20, anchors)
    inf.eval()
    x = synth.images(2, 96, seed=1)
    with torch.no_grad():
        pred = ns.model._inference(inf, x)
    np.savez_compressed(os.path.join(OUT, 'forward_narrow.npz'), **{k: v.numpy() for k, v in pred.items()})
    print('forward_narrow feature rms', pred['feature'].pow(2).mean().sqrt().item())

    # ---- 2. full-width Darknet-19, S=64 (2x2 grid), B=1, fp32 reference output + fp64 ground truth
    sd = odark.init_state_dict(5, 20, seed=0, head_scale=1 / 40.0)
    dnn, inf = build_ref_model(ns, sd, 20, anchors)
    inf.eval()
    x = synth.images(1, 64, seed=1)
    with torch.no_grad():
        pred = ns.model._inference(inf, x)
        f64 = dnn.double()(x.double())
    np.savez_compressed(os.path.join(OUT, 'forward_full.npz'), feature=pred['feature'].numpy(), feature_fp64=f64.numpy(),
                        iou=pred['iou'].numpy(), yx_min=pred['yx_min'].numpy(), yx_max=pred['yx_max'].numpy())
    print('forward_full feature rms', pred['feature'].pow(2).mean().sqrt().item(), 'fp32-fp64 max', (pred['feature'].double() - f64).abs().max().item())

    # ---- 3. decode alone on a 0.5*randn feature, 13x13, C=20 and C=80 / single-class
    for name, A, C, rows in (('decode_voc', 5, 20, 13), ('decode_coco', 5, 80, 10), ('decode_1cls', 5, 1, 13)):
        g = torch.Generator().manual_seed(7)
        feat = 0.5 * torch.randn(3, odark.output_channels(A, C), rows, rows, generator=g)

        class Id(torch.nn.Module):
            def forward(self, t):
                return t
        inf = ns.model.Inference(ref_config(), Id(), anchors[:A])
        with torch.no_grad():
            pred = ns.model._inference(inf, feat)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **{k: v.numpy() for k, v in pred.items()})

    # ---- 4. NMS known answers from the reference loop
    keeps = {}
    for n in (0, 1, 2, 50, 200, 845, 2000):
        score, mn, mx = synth.nms_boxes(n)
        for ov in (0.45, 0.5):
            k = ns.postprocess.nms(torch.from_numpy(score), torch.from_numpy(mn).view(-1, 2), torch.from_numpy(mx).view(-1, 2), ov)
            keeps['n%d_ov%d' % (n, int(ov * 100))] = np.array([int(i) for i in k], np.int64)
    np.savez_compressed(os.path.join(OUT, 'nms.npz'), **keeps)
    print('nms kept', {k: len(v) for k, v in keeps.items()})

    # ---- 5. filter + postprocess (detect.py:51-80), fix=0 and fix=1
    det = detect_functions(ns)
    d = np.load(os.path.join(OUT, 'decode_voc.npz'))
    out = {}
    for fix in (0, 1):
        cfg = det_config(fix)
        for b in range(d['iou'].shape[0]):
            iou = torch.from_numpy(d['iou'][b]).view(-1)
            mn = torch.from_numpy(d['yx_min'][b]).view(-1, 2)
            mx = torch.from_numpy(d['yx_max'][b]).view(-1, 2)
            prob = torch.softmax(torch.from_numpy(d['logits'][b]), -1).view(iou.numel(), -1)
            r = det['postprocess'](cfg, iou, mn, mx, prob)
            tag = 'fix%d_b%d_' % (fix, b)
            out[tag + 'none'] = np.array(r is None)
            if r is not None:
                for name, t in zip(('iou', 'yx_min', 'yx_max', 'cls', 'score'), r):
                    out[tag + name] = t.numpy()
    np.savez_compressed(os.path.join(OUT, 'postprocess.npz'), **out)

    # ---- 6. region loss (patched reference), B=2, 13x13, int labels (CE) and one-hot (MSE)
    mp = patched_loss_module(ns)
    res = {}
    for onehot in (False, True):
        g = torch.Generator().manual_seed(11)
        feat = (0.5 * torch.randn(2, 125, 13, 13, generator=g)).requires_grad_(True)

        class Id(torch.nn.Module):
            def forward(self, t):
                return t
        inf = mp.Inference(ref_config(), Id(), anchors)
        pred = mp._inference(inf, feat)
        data = synth.norm_data(synth.labels(2, 416, 20, seed=2, onehot=onehot), 416, 416, 13, 13)
        loss, debug = mp.loss(anchors, data, pred, 0.6)
        tot = sum(loss[k] * w for k, w in dict(foreground=5, background=1, center=1, size=1, cls=1).items())
        tot.backward()
        tag = 'onehot_' if onehot else 'ce_'
        for k, v in loss.items():
            res[tag + k] = v.detach().numpy()
        res[tag + 'grad'] = feat.grad.numpy()
        res[tag + 'positive'] = debug['positive'].numpy()
        res[tag + 'negative'] = debug['negative'].numpy()
        res[tag + 'best_iou'] = debug['iou'].numpy()
        print('loss', tag, {k: float(v) for k, v in loss.items()})
    np.savez_compressed(os.path.join(OUT, 'loss.npz'), **res)

    # ---- 7. the reference's own IoU known-answer tests must pass on the reference here
    import unittest
    suite = unittest.TestSuite()
    for cls in (ns.iou.TestIouMatrix, ns.iou.TestBatchIouMatrix, ns.iou.TestBatchIouPair):
        suite.addTests(unittest.defaultTestLoader.loadTestsFromTestCase(cls))
    r = unittest.TextTestRunner(verbosity=0).run(suite)
    assert r.wasSuccessful()
    print('golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
