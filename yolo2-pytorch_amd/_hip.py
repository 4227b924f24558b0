"""ctypes binding of libyolo2_hip.so (C ABI: include/yolo2_hip.h).

Plumbing only: torch provides device memory and the current HIP stream; every
numerical operation on the hot path happens inside the library.  There is NO
CPU fallback: if the library is missing, `lib()` raises, and every wrapper
refuses non-GPU tensors.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libyolo2_hip.so')
_lib = None

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class ConvParams(ctypes.Structure):
    """struct y2_conv_params (include/yolo2_hip.h)."""
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
                ('y', c_void_p), ('y_pool', c_void_p), ('stats', c_void_p),
                ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
                ('Cin', ctypes.c_int32), ('ldx', ctypes.c_int32), ('Cout', ctypes.c_int32), ('ksize', ctypes.c_int32),
                ('ldy', ctypes.c_int32), ('coff', ctypes.c_int32), ('ldp', ctypes.c_int32), ('poff', ctypes.c_int32),
                ('out_mode', ctypes.c_int32), ('slope', c_float), ('tile', ctypes.c_int32)]


# name -> argtypes; restype is int for everything except y2_build_info
SIGNATURES = {
    'y2_abi_version': [],
    'y2_pack_weight': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    'y2_unpack_weight_grad': [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'y2_bn_fold': [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p],
    'y2_conv_fwd': [ctypes.POINTER(ConvParams), c_void_p],
    'y2_conv_fwd_batch': [ctypes.POINTER(ConvParams), c_int, c_void_p],
    'y2_conv0_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    'y2_maxpool2_fwd': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_decode': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'y2_filter_visible': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'y2_iou_matrix': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    'y2_iou_pair': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p],
    'y2_conv_wgrad': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_conv0_wgrad': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_bn_finalize': [c_void_p, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    'y2_bn_act_fwd': [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_bn_act_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                      c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_colsum': [c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_void_p],
    'y2_f64_to_f32': [c_void_p, c_void_p, c_int, ctypes.c_double, c_void_p],
    'y2_decode_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    'y2_region_loss_fwd': [c_void_p] * 11 + [c_int] * 6 + [c_float] + [c_void_p] * 5 + [c_void_p],
    'y2_region_loss_finalize': [c_void_p, ctypes.c_double, c_int, c_void_p, c_void_p],
    'y2_region_loss_bwd': [c_void_p] * 9 + [c_int] * 6 + [c_float] + [c_void_p] * 5 + [c_void_p] * 4 + [c_void_p],
    'y2_nms': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
}

STATS_REPL = 32   # Y2_STATS_REPL (include/yolo2_hip.h)

ERRORS = {-1: 'Y2_EINVAL (bad size / null pointer)', -2: 'Y2_EALIGN (unaligned pointer or stride)', -3: 'Y2_ENOSUP (unsupported combination)'}


class HipLibraryMissing(RuntimeError):
    pass


def build(verbose=False):
    """Compile libyolo2_hip.so in-tree with hipcc for gfx950 (works without a GPU)."""
    out = subprocess.run(['bash', os.path.join(_HERE, 'csrc', 'build.sh')], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout + out.stderr)
    if out.returncode != 0:
        raise RuntimeError('building libyolo2_hip.so failed')
    return LIB_PATH


def lib():
    """The loaded library; raises HipLibraryMissing (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing('%s not found: run `python __graft_entry__.py build` (or yolo2-pytorch_amd/csrc/build.sh); '
                                    'the YOLOv2 hot path has no CPU/eager fallback' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        l.y2_build_info.argtypes = []
        l.y2_build_info.restype = ctypes.c_char_p
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        if rc <= -1000:
            raise RuntimeError('%s: HIP error %d' % (what, -rc - 1000))
        raise RuntimeError('%s: %s' % (what, ERRORS.get(rc, rc)))


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('yolo2-hip: this operator runs only on an MI355X GPU tensor (got device %s); there is no CPU fallback' % t.device)


def f32c(t):
    """Contiguous fp32 view/copy of a GPU tensor."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
