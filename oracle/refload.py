"""Load the reference's hot-path modules BY FILE PATH behind a tiny `utils` shim.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only where
/root/reference exists (the build container); nothing on the GPU box imports
this.  It is used by oracle/make_golden.py to pin the oracle restatement and to
produce the committed fixtures under tests/golden/.

Why a shim: the reference's `utils/__init__.py:109` uses `async` as a
parameter name (SyntaxError on Python >= 3.7), so the package itself cannot be
imported; the hot-path files only need `utils.ensure_device` and
`utils.iou.torch` from it (model/__init__.py:25, utils/postprocess.py:21).
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get('YOLO2_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'model'))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load():
    """Returns a namespace with the reference modules: iou, postprocess, model, yolo2.

    The modules are registered under private names (``_ref_*``) except for the
    `utils`/`model` names the reference files import themselves; those are
    swapped in only for the duration of the load and then removed again so the
    repo's own `utils`/`model` packages are never shadowed.
    """
    saved = {k: sys.modules.get(k) for k in ('utils', 'utils.iou', 'utils.iou.torch', 'utils.postprocess', 'model', 'model.yolo2')}
    for k in saved:
        sys.modules.pop(k, None)
    try:
        utils = types.ModuleType('utils')
        utils.__path__ = []

        def ensure_device(t, device_id=None, non_blocking=False):
            # reference utils/__init__.py:109-112 — CPU-only here
            return t
        utils.ensure_device = ensure_device
        sys.modules['utils'] = utils
        iou_pkg = types.ModuleType('utils.iou')
        iou_pkg.__path__ = []
        sys.modules['utils.iou'] = iou_pkg
        utils.iou = iou_pkg
        iou_torch = _load('utils.iou.torch', os.path.join(REF, 'utils/iou/torch.py'))
        iou_pkg.torch = iou_torch
        post = _load('utils.postprocess', os.path.join(REF, 'utils/postprocess.py'))
        utils.postprocess = post
        model = _load('model', os.path.join(REF, 'model/__init__.py'))
        yolo2 = _load('model.yolo2', os.path.join(REF, 'model/yolo2.py'))
        ns = types.SimpleNamespace(iou=iou_torch, postprocess=post, model=model, yolo2=yolo2, utils=utils)
    finally:
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    return ns
