"""tests/golden/darknet_weights.npz from the REFERENCE's own convert_darknet_torch.main() and checksum_torch.main().

    python -m oracle.make_golden_darknet_weights          # build container only (needs /root/reference)

Both scripts are exec'd from their files with the imports they cannot satisfy here replaced by in-memory stand-ins (never
written back, never copied): `humanize`, `yaml` + logging.config (logging set-up only), the `utils` package (its
__init__ does not import on Python >= 3.7: the helpers main() calls are re-provided with the reference's semantics and the
REAL utils.abs_mean formula), `utils.train.Saver` / `load_model` (capture / hand back the state_dict instead of touching
a model directory), `transform`, `cv2` (imported but unused by the path).  `model` / `model.yolo2` are the reference's own files
(oracle/refload.py).  Two torch-0.3 / numpy-1 idioms are patched in memory: ndarray.tostring() -> tobytes(),
Variable(tensor, volatile=True) -> the tensor under no_grad.  The network is the reference Darknet with ratio=1/32 (the
constructor's own width multiplier, model/yolo2.py:69-72) so that the synthetic file is ~200 KB instead of 203 MB.
"""
import collections
import contextlib
import functools
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import darknet_weights as odw  # noqa: E402
from oracle import refload  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'darknet_weights.npz')
RATIO = 1.0 / 32
NUM_CLS = 20
FILE_SEED = 41
HEADER = (0, 1, 0, 12800)
TRAILING = 12           # bytes left over at the end of the file ("%d bytes remaining" warning path, :116-117)


def synthetic_file(shapes):
    rng = np.random.RandomState(FILE_SEED)
    arrays = []
    for key, shape in odw.file_order(shapes):
        a = rng.standard_normal(shape).astype(np.float32) * 0.5
        if key.endswith('running_var'):
            a = np.abs(a) + 0.5
        arrays.append(a)
    return odw.write_weights(arrays, HEADER) + b'\x00' * TRAILING


def run_reference(weights_path):
    ns = refload.load()
    captured = {}

    class Saver(object):
        def __init__(self, model_dir, keep, logger=None):
            self.ext = '.pth'

        def __call__(self, state_dict, step, epoch):
            captured['state_dict'] = state_dict
            return os.path.join(os.path.dirname(weights_path), 'model')

    anchors_tsv = np.loadtxt(os.path.join(refload.REF, 'config/anchors/voc.tsv'), skiprows=1, dtype=np.float32)
    utils = types.ModuleType('utils')
    utils.__path__ = []
    utils.load_config = lambda config, paths: config.read(os.path.join(refload.REF, 'config.ini'))
    utils.modify_config = lambda config, cmd: None
    utils.get_cache_dir = lambda config: '/nonexistent'
    utils.get_model_dir = lambda config: '/nonexistent'
    utils.get_category = lambda config, cache_dir=None: ['c%d' % i for i in range(NUM_CLS)]
    utils.get_anchors = lambda config, dtype=np.float32: anchors_tsv[:, [1, 0]].copy()        # utils/__init__.py:78-81: (height, width)
    utils.parse_attr = lambda s: functools.partial(ns.yolo2.Darknet, ratio=RATIO)
    utils.abs_mean = lambda data, dtype=np.float32: np.sum(np.abs(data)) / dtype(data.size)   # utils/__init__.py:119-121 verbatim semantics
    utils_train = types.ModuleType('utils.train')
    utils_train.Saver = Saver
    utils_train.load_model = lambda model_dir: ('captured', 0, 0)
    utils.train = utils_train
    humanize = types.ModuleType('humanize')
    humanize.naturalsize = lambda n: '%d B' % n
    yaml = types.ModuleType('yaml')
    yaml.load = lambda f: {'version': 1}
    transform = types.ModuleType('transform')
    cv2 = types.ModuleType('cv2')         # imported by checksum_torch.py:28, never used on this path
    stubs = {'utils': utils, 'utils.train': utils_train, 'humanize': humanize, 'yaml': yaml, 'transform': transform, 'cv2': cv2,
             'model': ns.model, 'model.yolo2': ns.yolo2}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    real_load = torch.load
    logging_yml = os.path.join(os.path.dirname(weights_path), 'logging.yml')
    open(logging_yml, 'w').write('version: 1\n')
    argv = sys.argv
    try:
        # ---- convert_darknet_torch.main()
        src = open(os.path.join(refload.REF, 'convert_darknet_torch.py')).read().replace('.tostring()', '.tobytes()')
        g = {'__name__': 'ref_convert_darknet_torch'}
        exec(compile(src, 'convert_darknet_torch.py', 'exec'), g)
        sys.argv = ['convert_darknet_torch.py', weights_path, '--logging', logging_yml]
        g['main']()
        state_dict = captured['state_dict']
        # ---- checksum_torch.main() on the converted state_dict
        torch.load = lambda path, map_location=None: state_dict
        src = open(os.path.join(refload.REF, 'checksum_torch.py')).read().replace('.tostring()', '.tobytes()')
        src = src.replace('output = dnn(torch.autograd.Variable(tensor, volatile=True)).data', 'with torch.no_grad():\n        output = dnn.eval()(tensor)')
        g2 = {'__name__': 'ref_checksum_torch'}
        exec(compile(src, 'checksum_torch.py', 'exec'), g2)
        sys.argv = ['checksum_torch.py', '--logging', logging_yml, '-s', '0']
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g2['main']()
        return state_dict, buf.getvalue()
    finally:
        sys.argv = argv
        torch.load = real_load
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def reference_shapes():
    import configparser
    ns = refload.load()
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(refload.REF, 'config.ini'))
    anchors = torch.zeros(5, 2)
    dnn = ns.yolo2.Darknet(ns.model.ConfigChannels(cfg), anchors, NUM_CLS, ratio=RATIO)
    return collections.OrderedDict((k, tuple(v.shape)) for k, v in dnn.state_dict().items() if not k.endswith('num_batches_tracked'))


def main():
    import tempfile
    assert refload.available(), 'needs /root/reference'
    shapes = reference_shapes()
    data = synthetic_file(shapes)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'synthetic.weights')
        open(path, 'wb').write(data)
        state_dict, checksum_out = run_reference(path)
    # the restatement must agree with what the reference produced
    mine, header, remaining = odw.read_weights(data, shapes, 5)
    assert header == HEADER and remaining == TRAILING
    assert list(mine.keys()) == list(state_dict.keys())
    for k in mine:
        assert np.array_equal(mine[k], state_dict[k].numpy()), k
    rows = [r for r in checksum_out.splitlines() if r.strip()]
    for r in rows:
        key = r.split('\t')[0]
        if key in mine:
            assert r == odw.checksum_row(key, mine[key]), (r, odw.checksum_row(key, mine[key]))
    out = {'sd/' + k: v.numpy() for k, v in state_dict.items()}
    out['keys'] = np.array(list(state_dict.keys()))
    out['shape_keys'] = np.array(list(shapes.keys()))
    out['shape_vals'] = np.array([','.join(map(str, s)) for s in shapes.values()])
    out['checksum_rows'] = np.array(rows)
    out['file_sha'] = np.array(__import__('hashlib').sha256(data).hexdigest())
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, '%d keys, file %d bytes, %d checksum rows' % (len(state_dict), len(data), len(rows)))
    print('\n'.join(rows[-3:]))


if __name__ == '__main__':
    main()
