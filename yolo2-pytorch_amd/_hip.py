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
LIB_PATH = os.environ.get('Y2_LIB') or os.path.join(_HERE, 'csrc', 'libyolo2_hip.so')     # Y2_LIB: A/B builds of the same ABI (tools/)
_lib = None

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class ConvParams(ctypes.Structure):
    """struct y2_conv_params (include/yolo2_hip.h)."""
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
                ('y', c_void_p), ('y_pool', c_void_p), ('stats', c_void_p),
                ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
                ('Cin', ctypes.c_int32), ('ldx', ctypes.c_int32), ('Cout', ctypes.c_int32), ('ksize', ctypes.c_int32),
                ('ldy', ctypes.c_int32), ('coff', ctypes.c_int32), ('ldp', ctypes.c_int32), ('poff', ctypes.c_int32),
                ('out_mode', ctypes.c_int32), ('slope', c_float), ('tile', ctypes.c_int32),
                ('workspace', c_void_p), ('workspace_bytes', ctypes.c_int64),
                ('residual', c_void_p), ('ldr', ctypes.c_int32), ('stride', ctypes.c_int32), ('pad_plus1', ctypes.c_int32),
                ('transposed', ctypes.c_int32), ('out_h', ctypes.c_int32), ('out_w', ctypes.c_int32), ('algo', ctypes.c_int32),
                ('w_plane', ctypes.c_int64)]


# name -> argtypes; restype is int for everything except y2_build_info
class OptTensor(ctypes.Structure):
    """y2_opt_tensor (include/yolo2_hip.h)."""
    _fields_ = [('param', c_void_p), ('grad', c_void_p), ('state1', c_void_p), ('state2', c_void_p), ('numel', ctypes.c_int64)]


OPT_MAX_TENSORS = 48


class PrepItem(ctypes.Structure):
    """y2_prep_item (include/yolo2_hip.h)."""
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('Cout', ctypes.c_int32), ('Cin', ctypes.c_int32), ('ksize', ctypes.c_int32), ('mode', ctypes.c_int32)]


PREP_FPROP, PREP_DGRAD, PREP_WINO_FPROP, PREP_WINO_DGRAD, PREP_WINO6_DGRAD = 0, 1, 2, 3, 4


class MultiItem(ctypes.Structure):
    """y2_multi_item (include/yolo2_hip.h)."""
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('n', ctypes.c_int64), ('op', ctypes.c_int32), ('reserved', ctypes.c_int32)]


MULTI_ZERO, MULTI_F64_TO_F32, MULTI_COPY = 0, 1, 2

SIGNATURES = {
    'y2_abi_version': [],
    'y2_pack_weight': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    'y2_prep_weights': [ctypes.POINTER(PrepItem), c_int, c_void_p],
    'y2_expand_classes': [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_float] + [c_void_p] * 8 + [c_void_p],
    'y2_iou_rowmax': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    'y2_split_bf16x3': [c_void_p, c_void_p, ctypes.c_longlong, c_void_p],
    'y2_split_f16x2': [c_void_p, c_void_p, ctypes.c_longlong, c_float, c_void_p],
    'y2_split_f16_overflow': [c_int],
    'y2_gemm_split_f16': [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int, c_float, c_void_p],
    'y2_gemm_split': [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int, c_void_p],
    'y2_multi': [ctypes.POINTER(MultiItem), c_int, c_void_p],
    'y2_small_dot': [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    'y2_small_scale': [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    'y2_unpack_weight_grad': [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'y2_bn_fold': [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p],
    'y2_conv_fwd': [ctypes.POINTER(ConvParams), c_void_p],
    'y2_conv_fwd_workspace_bytes': [ctypes.POINTER(ConvParams)],
    'y2_wino_weight': [c_void_p, c_void_p, c_int, c_int, c_void_p],
    'y2_wino6_weight': [c_void_p, c_void_p, c_int, c_int, c_void_p],
    'y2_opt_sgd': [ctypes.POINTER(OptTensor), c_int, c_float, c_float, c_float, c_float, c_int, c_int, c_void_p],
    'y2_opt_adam': [ctypes.POINTER(OptTensor), c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p],
    'y2_opt_grad_sumsq': [ctypes.POINTER(OptTensor), c_int, c_void_p, c_void_p],
    'y2_opt_clip_grads': [ctypes.POINTER(OptTensor), c_int, c_void_p, c_float, c_void_p],
    'y2_wino_wgrad_workspace_bytes': [c_int, c_int, c_int, c_int, c_int],
    'y2_wino_wgrad_workspace_bytes_ex': [c_int, c_int, c_int, c_int, c_int, c_int],
    'y2_wino_wgrad': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p],
    'y2_wino_wgrad_ex': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p],
    'y2_conv_fwd_batch': [ctypes.POINTER(ConvParams), c_int, c_void_p],
    'y2_conv0_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    'y2_maxpool2_fwd': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_maxpool_fwd': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_nchw_to_nhwc': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_decode': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'y2_filter_visible': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    'y2_iou_matrix': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    'y2_iou_pair': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p],
    'y2_conv_wgrad': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_conv_wgrad_ex': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_conv0_wgrad_fused': [c_void_p] * 7 + [c_float, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    'y2_conv0_wgrad': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_bn_finalize': [c_void_p, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    'y2_bn_act_fwd': [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_bn_act_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                      c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_bn_act_fwd_ex': [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_bn_act_bwd_ex': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                         c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_wino6_tiles': [c_int, c_int, c_int],
    'y2_bn_act_bwd_wino6': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                            c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_maxpool_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    'y2_colsum': [c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_void_p],
    'y2_f64_to_f32': [c_void_p, c_void_p, c_int, ctypes.c_double, c_void_p],
    'y2_decode_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    'y2_region_loss_fwd': [c_void_p] * 11 + [c_int] * 6 + [c_float] + [c_void_p] * 5 + [c_void_p],
    'y2_region_loss_finalize': [c_void_p, ctypes.c_double, c_int, c_void_p, c_void_p],
    'y2_region_loss_bwd': [c_void_p] * 9 + [c_int] * 6 + [c_float] + [c_void_p] * 5 + [c_void_p] * 4 + [c_void_p],
    'y2_nms_host': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    'y2_iou_matrix_host': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p],
    'y2_iou_pair_host': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p],
    'y2_set_deterministic': [c_int, c_void_p, ctypes.c_longlong],
    'y2_get_deterministic': [],
    'y2_colstats_det': [c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p],
    'y2_prof_enable': [c_int],
    'y2_prof_count': [],
    'y2_prof_set_tag': [c_int],
    'y2_prof_get_tag': [c_int],
    'y2_prof_get': [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_float), ctypes.POINTER(ctypes.c_double)],
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
        l.y2_conv_fwd_workspace_bytes.restype = ctypes.c_longlong
        l.y2_wino_wgrad_workspace_bytes.restype = ctypes.c_longlong
        l.y2_wino6_tiles.restype = ctypes.c_longlong
        l.y2_wino_wgrad_workspace_bytes_ex.restype = ctypes.c_longlong
        l.y2_build_info.argtypes = []
        l.y2_build_info.restype = ctypes.c_char_p
        _lib = l
    return _lib


LAUNCHES = [0]      # library entry points that returned through check(): what a capture segment counts to know it is not empty


def check(rc, what):
    LAUNCHES[0] += 1
    if rc != 0:
        if rc <= -1000:
            raise RuntimeError('%s: HIP error %d' % (what, -rc - 1000))
        raise RuntimeError('%s: %s' % (what, ERRORS.get(rc, rc)))


_LAUNCH = None      # a torch.cuda.Stream the library's launches go to instead of torch's current stream (launch_on)


def stream():
    return c_void_p((_LAUNCH if _LAUNCH is not None else torch.cuda.current_stream()).cuda_stream)


class launch_on(object):
    """Library launches inside the block go to stream `s` while torch's CURRENT stream - the one the caching allocator tags new blocks with - stays what
    it is.  The training backward forks its weight gradients this way: their kernels run on a side stream, what they write is allocated on the main
    stream like everything else of the step, so a step's memory has ONE allocation stream (a freed block is reusable at once, by the next plan too: no
    record_stream, no per-stream free lists in the step's arena).  The caller orders the two streams with events and keeps what the side stream reads
    alive until the main stream has waited for it."""

    def __init__(self, s):
        self.s, self.prev = s, None

    def __enter__(self):
        global _LAUNCH
        self.prev, _LAUNCH = _LAUNCH, self.s
        return self.s

    def __exit__(self, *exc):
        global _LAUNCH
        _LAUNCH = self.prev
        return False


def _timing_stream():
    """Context for a block that TIMES launches with torch events: the events must sit on the stream the launches go to."""
    import contextlib
    return torch.cuda.stream(_LAUNCH) if _LAUNCH is not None else contextlib.nullcontext()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def multi(items, st=None):
    """One y2_multi launch.  items: (op, dst tensor, src tensor or None) - MULTI_ZERO fills fp32 / fp64 / integer tensors with zero
    bytes (the element count is converted to fp32 words), the other ops take the element count of dst."""
    if not items:
        return
    table = (MultiItem * len(items))()
    for e, (op, dst, src) in zip(table, items):
        e.op, e.dst = op, dst.data_ptr()
        if op == MULTI_ZERO:
            nbytes = dst.numel() * dst.element_size()
            assert nbytes % 4 == 0 and dst.is_contiguous()
            e.n, e.src = nbytes // 4, None
        else:
            assert dst.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous() and src.numel() >= dst.numel()
            e.n, e.src = dst.numel(), src.data_ptr()
    check(lib().y2_multi(table, len(items), stream() if st is None else st), 'y2_multi')


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('yolo2-hip: this operator runs only on an MI355X GPU tensor (got device %s); there is no CPU fallback' % t.device)


# Memory written through raw pointers (the fused optimizers, y2_bn_finalize's running statistics) is invisible to torch's
# in-place bookkeeping, so every such writer reports the tensors it wrote with `wrote(...)`: their `_version` counters advance
# exactly as if a torch in-place op had run.  Caches of packed / folded / transformed weights key on (data_ptr, _version) of
# the tensors they were derived from and nothing else: per tensor, per model - training one model never invalidates another
# model's caches or captured graphs.
def wrote(tensors):
    """Advance torch's version counter of every tensor in `tensors` (Parameters / buffers written by a HIP kernel)."""
    try:
        torch.autograd.graph.increment_version(tensors)
    except TypeError:          # torch releases whose increment_version takes ONE tensor (README: developed on torch 2.10)
        for t in tensors:
            torch.autograd.graph.increment_version(t)


# Execution plans additionally depend on the measured algorithm table: adopting another process's table (import_tune) moves this.
_TUNE_EPOCH = [0]


def tune_epoch():
    return _TUNE_EPOCH[0]


# ---- deterministic mode (include/yolo2_hip.h: y2_set_deterministic): fixed-order reductions instead of atomics, heuristic instead of
# timed algorithm selection.  Y2_DETERMINISTIC=1 turns it on at the first GPU use; set_deterministic() switches at run time.
DETERMINISTIC = os.environ.get('Y2_DETERMINISTIC', '0') == '1'
_DET_WS = {}
_DET_BYTES = int(os.environ.get('Y2_DET_WS_MB', '256')) << 20


def set_deterministic(on, dev=None):
    """Bit-reproducible training on `dev` (default: the current GPU): the library's reductions go through a scratch area owned here."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    L = lib()
    if not on:
        check(L.y2_set_deterministic(0, None, 0), 'y2_set_deterministic')
        return
    dev = torch.device('cuda', torch.cuda.current_device()) if dev is None else torch.device(dev)
    ws = _DET_WS.get(str(dev))
    if ws is None:
        ws = torch.empty(_DET_BYTES // 4, dtype=torch.float32, device=dev)
        _DET_WS[str(dev)] = ws
    check(L.y2_set_deterministic(1, ws.data_ptr(), ws.numel() * 4), 'y2_set_deterministic')


def ensure_deterministic(dev):
    """Called by the training graph: arm the library when the mode was requested (env / set_deterministic) but not yet armed on this device."""
    if DETERMINISTIC and (str(torch.device(dev)) not in _DET_WS or not lib().y2_get_deterministic()):
        set_deterministic(True, dev)
    return DETERMINISTIC


def colstats_det(z, M, C, ld, stats):
    """Deterministic BatchNorm statistics of the raw convolution output into copy 0 of a zeroed stats buffer (y2_colstats_det)."""
    G = min(1024, (M + 255) // 256)
    key = 'colstats:' + str(z.device)
    ws = _DET_WS.get(key)
    if ws is None or ws.numel() * 4 < G * 2 * C * 8:
        ws = torch.empty(max(G * 2 * C * 2, 1 << 18), dtype=torch.float32, device=z.device)
        _DET_WS[key] = ws
    check(lib().y2_colstats_det(ptr(z), M, C, ld, ptr(stats), ptr(ws), ws.numel() * 4, stream()), 'y2_colstats_det')


_WS = {}
# A hipGraph bakes in the addresses of every scratch buffer its kernels were launched with.  The process-wide scratch caches
# (_WS, _WGRAD_WS) grow by REPLACING their tensor, which would leave a captured graph writing into memory the allocator has handed to
# somebody else - so while a training step is being captured (model.train_graph.StepPlan) SCOPE points at a dict owned by that plan and
# every cache lookup below goes there instead: the plan's scratch lives exactly as long as its graphs.
SCOPE = None


def _cache(d):
    if SCOPE is None:
        return d
    return SCOPE.setdefault(('_hip', id(d)), {})


def workspace(dev, nbytes):
    """Per-device scratch for the conv split-K remainder scheme (grows on demand; consecutive convolutions on one stream
    may share it: a layer's fix-up kernel has consumed it before the next layer's main kernel starts)."""
    key = str(dev)
    ws = _cache(_WS)
    t = ws.get(key)
    if t is None or t.numel() * 4 < nbytes:
        _retire(t)
        t = torch.empty(max(int(nbytes) // 4 + 4, 1024), dtype=torch.float32, device=dev)
        ws[key] = t
    return t


def _retire(t):
    """A scratch buffer is being replaced by a larger one while a capture scope is active: graphs captured earlier into the same scope (the plans of other
    input sizes share one, train.StepRunner) hold the OLD address - it stays alive with the scope."""
    if t is not None and SCOPE is not None:
        SCOPE.setdefault(('_hip', 'retired'), []).append(t)


def conv_workspace(params, dev):
    """Attach scratch to one ConvParams (query + cached allocation)."""
    need = lib().y2_conv_fwd_workspace_bytes(ctypes.byref(params))
    if need > 0:
        ws = workspace(dev, need)
        params.workspace, params.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    else:
        params.workspace, params.workspace_bytes = None, 0
    return need


class PlanCache(object):
    """Small LRU of execution plans keyed on the input shape (multi-scale training / evaluation cycles through ~10 sizes,
    utils/data.py:135-141: a single-entry cache would re-plan, re-allocate and re-tune at every size switch).  Bounded by entry
    count and by the bytes of the intermediate buffers the plans own (Y2_PLAN_CACHE / Y2_PLAN_CACHE_GB)."""

    def __init__(self, entries=None, gbytes=None):
        import collections
        self.entries = int(os.environ.get('Y2_PLAN_CACHE', '12')) if entries is None else entries
        self.bytes = int(float(os.environ.get('Y2_PLAN_CACHE_GB', '64')) * (1 << 30)) if gbytes is None else int(gbytes * (1 << 30))
        self.d = collections.OrderedDict()
        self.hits = self.misses = 0

    def get(self, key):
        plan = self.d.get(key)
        if plan is None:
            self.misses += 1
            return None
        self.hits += 1
        self.d.move_to_end(key)
        return plan

    def put(self, key, plan, nbytes):
        self.d[key] = plan
        plan['nbytes'] = nbytes
        while len(self.d) > 1 and (len(self.d) > self.entries or sum(p['nbytes'] for p in self.d.values()) > self.bytes):
            self.d.popitem(last=False)

    def latest(self):
        return next(reversed(self.d.values())) if self.d else None

    def clear(self):
        self.d.clear()


_TUNE = {}
TUNE_MISSES = []          # keys measured in this process (not served by Y2_TUNE_CACHE / the committed default table)
AUTOTUNE = os.environ.get('Y2_AUTOTUNE', '1') != '0'
TUNE_CACHE = os.environ.get('Y2_TUNE_CACHE')      # optional JSON file persisting the measured tile choices across processes
if TUNE_CACHE and os.path.exists(TUNE_CACHE):
    try:
        import json as _json
        _TUNE.update({tuple(_json.loads(k)): v for k, v in _json.load(open(TUNE_CACHE)).items()})   # v = [algo, tile]
    except Exception:
        pass


# ---- committed default table: the per-layer choices measured on an MI355X for the BASELINE shapes (tools/make_tune_table.py), so that a
# fresh box - and each of the eight ranks of a node - starts on the measured plans instead of spending 1.4-6 s per new input shape on
# timing runs.  It is valid for the kernels it was measured on: the file carries a hash of csrc/*.hip + headers and is ignored when the
# sources differ.  Entries never override what this process measured or loaded from Y2_TUNE_CACHE.  Y2_TUNE_DEFAULTS=0: do not load.
TUNE_DEFAULTS = os.environ.get('Y2_TUNE_DEFAULTS', '1') != '0'
DEFAULTS_PATH = os.path.join(_HERE, 'tune', 'default_gfx950.json')
_DEFAULTS_SEEN = {}


def kernel_hash():
    """sha256 (16 hex digits) of the kernel sources the library is built from; None when the sources are not there."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.hip'))) + [os.path.join(_HERE, 'csrc', 'common.h'), os.path.join(os.path.dirname(_HERE), 'include', 'yolo2_hip.h')]
    if len(files) < 3 or not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        for line in open(f, 'r', errors='replace'):
            if line.startswith('// y2-build-flags:'):          # per-file code generation options (csrc/build.sh) are code
                h.update(line.strip().encode())
                h.update(b'\n')
            # code only: a // comment (outside a string literal) and blank lines do not make measured choices or profiles stale
            cut = line.find('//')
            while cut >= 0 and line.count('"', 0, cut) % 2:
                cut = line.find('//', cut + 2)
            code = (line if cut < 0 else line[:cut]).strip()
            if code:
                h.update(code.encode())
                h.update(b'\n')
    return h.hexdigest()[:16]


def load_tune_defaults(dev, path=None):
    """Adopt the committed default choices for device `dev` (once per device).  Returns the number of entries adopted."""
    key = str(dev)
    if key in _DEFAULTS_SEEN and path is None:
        return _DEFAULTS_SEEN[key]
    n = 0
    path = path or DEFAULTS_PATH
    if TUNE_DEFAULTS and os.path.exists(path):
        try:
            import json as _json
            d = _json.load(open(path))
            if d.get('kernels') is not None and (d.get('kernels') == kernel_hash() or os.environ.get('Y2_TUNE_STALE_OK') == '1'):      # (the override: development runs between a kernel edit and the table's regeneration)
                for k, v in d['entries']:
                    kk = tuple(key if e == '@dev' else e for e in k)
                    if kk not in _TUNE:
                        _TUNE[kk] = v
                        n += 1
        except Exception:
            n = 0
    _DEFAULTS_SEEN[key] = n
    return n


def save_tune_defaults(path=None, note=''):
    """tools/make_tune_table.py: write this process's measured choices as the default table for the current kernel sources."""
    import json as _json
    path = path or DEFAULTS_PATH
    os.makedirs(os.path.dirname(path), exist_ok=True)
    entries = [[list(k), v] for k, v in export_tune()]
    with open(path, 'w') as f:          # one entry per line: diffs of a regenerated table stay readable
        f.write('{"kernels": %s, "note": %s, "entries": [\n' % (_json.dumps(kernel_hash()), _json.dumps(note)))
        f.write(',\n'.join(_json.dumps(e) for e in entries))
        f.write('\n]}\n')
    return len(entries)


WINOGRAD = os.environ.get('Y2_WINOGRAD', '1') != '0'     # 0: never pick the Winograd F(2x2,3x3) algorithm
WGRAD_F34 = os.environ.get('Y2_WGRAD_F34', '1') != '0'     # offer the 4x4-tile Winograd weight gradient (F(3x3, 4x4)) to the per-layer measurement
FORCE_ALGO = os.environ.get('Y2_FORCE_ALGO') or None     # 'direct' | 'winograd' | 'fused' | 'implicit' | 'fused3' | 'implicit3' | 'split': no autotune, that algorithm wherever the library accepts it
if FORCE_ALGO not in (None, 'direct', 'winograd', 'fused', 'implicit', 'fused3', 'implicit3', 'split'):
    raise ValueError('Y2_FORCE_ALGO must be direct, winograd, fused, implicit, fused3, implicit3 or split (got %r)' % FORCE_ALGO)      # ('split': the algorithm of the current split mode; '...3': the two-workgroups-per-CU kernel)
# Opt-in precision modes: the Winograd GEMMs may run on the bf16 / fp16 matrix pipe from split operands (csrc/gemm_split.hip): fp32-level
# accuracy (the same parity tests run in these modes).  SPLIT = 'bf16' (True): three bf16 planes per fp32 operand, six plane products
# (Y2_ALGO_WINOGRAD_SPLIT; fp32's exponent range); 'f16': two fp16 planes, three products, operands scaled by fixed powers of two
# (Y2_ALGO_WINOGRAD_SPLIT_F16; finite for |transformed activation| < 10^6 and |transformed weight| < 255 only).
SPLIT = 'f16' if os.environ.get('Y2_SPLIT_F16', '0') == '1' else ('bf16' if (os.environ.get('Y2_SPLIT_BF16', '0') == '1' or FORCE_ALGO == 'split') else '')
F16_U_SCALE = 256.0


def split_mode():
    """None, 'bf16' or 'f16' (SPLIT may be set to True by callers: the bf16 mode)."""
    return None if not SPLIT else ('f16' if SPLIT == 'f16' else 'bf16')


def split_overflowed(reset=True):
    """True when an operand of the fp16 split mode left fp16's range since the last reset (y2_split_f16_overflow; synchronises)."""
    rc = lib().y2_split_f16_overflow(int(bool(reset)))
    if rc < 0:
        check(rc, 'y2_split_f16_overflow')
    return rc == 1


def split_algo():
    return 5 if split_mode() == 'f16' else 4


def split_planes(t, mode=None):
    """The plane tuple of a contiguous fp32 GPU tensor for the current (or given) split mode: y2_split_bf16x3 -> [3][numel] bf16 (hi, mid, lo),
    y2_split_f16x2 -> [2][numel] fp16 of t * 256 (the weight-operand scale of Y2_ALGO_WINOGRAD_SPLIT_F16)."""
    require_gpu(t)
    t = f32c(t)
    if (mode or split_mode()) == 'f16':
        out = torch.empty(2 * t.numel(), dtype=torch.float16, device=t.device)
        check(lib().y2_split_f16x2(ptr(t), ptr(out), t.numel(), F16_U_SCALE, stream()), 'y2_split_f16x2')
        return out
    out = torch.empty(3 * t.numel(), dtype=torch.bfloat16, device=t.device)
    check(lib().y2_split_bf16x3(ptr(t), ptr(out), t.numel(), stream()), 'y2_split_bf16x3')
    return out


# Pinned gradient algorithms (tests: the training twin of Y2_FORCE_ALGO): every eligible data gradient on Winograd F(4x4,3x3) ('f43') or the
# direct kernel ('direct'); every eligible weight gradient on the direct kernel / the 2x2-tile / the 4x4-tile Winograd reduction.
FORCE_GRAD = os.environ.get('Y2_FORCE_GRAD_ALGO') or None
FORCE_WGRAD = os.environ.get('Y2_FORCE_WGRAD') or None
if FORCE_GRAD not in (None, 'f43', 'direct') or FORCE_WGRAD not in (None, 'direct', 'wino', 'f34'):
    raise ValueError('Y2_FORCE_GRAD_ALGO must be f43 or direct, Y2_FORCE_WGRAD direct, wino or f34')
PERSIST = os.environ.get('Y2_CONV_PERSIST', '1') != '0'      # 0: never offer the persistent-workgroup tiles (11 / 12 / 13 / 15) of the direct kernel (A/B runs)
IMPLICIT = os.environ.get('Y2_WINO_IMPLICIT', '1') != '0'  # 0: never offer Y2_ALGO_WINOGRAD_IMPLICIT (A/B runs)
WINO_MIN_CIN = 32                                        # below this the transforms cost more than the GEMM saves (measured; 32: the 208x208 layer, one K slab per tile of the fused kernels)


def wino_eligible(cout, cin, k, stride=1):
    """Shapes y2_conv_fwd accepts with algo = Y2_ALGO_WINOGRAD and where it can pay off."""
    return WINOGRAD and k == 3 and stride in (0, 1) and cin % 4 == 0 and cout % 4 == 0 and cin >= WINO_MIN_CIN


def wino6_weight(wp, cout, cin):
    """U6 [36][Cout][Cin] from a packed 3x3 weight: the filter operand of Y2_ALGO_WINOGRAD_F43 (4x4 output tiles; gradients only)."""
    u = torch.empty(36 * cout * cin, dtype=torch.float32, device=wp.device)
    check(lib().y2_wino6_weight(ptr(wp), ptr(u), cout, cin, stream()), 'y2_wino6_weight')
    return u


def wino_weight(wp, cout, cin):
    """U [16][Cout][Cin] from a packed 3x3 weight (y2_pack_weight mode 0 or 1)."""
    u = torch.empty(16 * cout * cin, dtype=torch.float32, device=wp.device)
    check(lib().y2_wino_weight(ptr(wp), ptr(u), cout, cin, stream()), 'y2_wino_weight')
    return u


def _time_conv(L, params, st):
    t = float('inf')
    for _ in range(3):          # best of three batches of 3 launches (DVFS / neighbour noise)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            L.y2_conv_fwd(ctypes.byref(params), st)
        e1.record()
        e1.synchronize()
        t = min(t, e0.elapsed_time(e1))
    return t


class OperandMissing(RuntimeError):
    """The chosen algorithm reads a filter operand the caller did not prepare (a captured training step derives only the forms its
    warm-up passes used: model.train_graph)."""


def autotune_conv(params, dev, wino_w=None, implicit_ok=True, wino_split=None, split_plane=0, f43=None, wino_eligible=None, peek=False):
    """Measure-don't-guess algorithm + tile selection for one y2_conv_fwd problem: the first time a problem shape is seen,
    every tile configuration of the direct kernel - and, when `wino_w` (y2_wino_weight output) is given, of the Winograd
    path - is timed (HIP events, best of 2 x 3 launches) and the fastest is cached for the process; later calls only
    look the answer up.  Sets params.algo / params.tile / params.w.  The outputs written while timing are real outputs.
    Never measures while a hipGraph is being captured (plans are built during warm-up).
    implicit_ok=False: the caller wants the transformed input left in the workspace (training keeps it for the weight gradient),
    so the algorithm that never materialises it (3) is not offered.
    f43: the result is a GRADIENT and may take Winograd F(4x4,3x3) (Y2_ALGO_WINOGRAD_F43, 8-9e-6 x rms per layer): the y2_wino6_weight operand
    or a callable producing it (called only when that algorithm is timed or chosen)."""
    # wino_eligible: the layer IS Winograd-eligible although `wino_w` is not at hand (None) - the problem keeps its identity (the key below),
    # and an algorithm that reads the missing operand raises OperandMissing instead of being silently replaced by another one
    wino_ok = (((wino_w is not None) if wino_eligible is None else bool(wino_eligible)) and WINOGRAD and params.ksize == 3 and params.stride in (0, 1)
               and params.pad_plus1 in (0, 2) and not params.transposed and not params.residual and params.out_mode == 0)
    split_ok = bool(wino_ok and SPLIT and wino_split is not None and params.Cin % 32 == 0)
    key = (params.B, params.H, params.W, params.Cin, params.ldx, params.Cout, params.ksize, bool(params.y), bool(params.y_pool),
           bool(params.stats), params.out_mode, params.stride, params.pad_plus1, bool(params.residual), params.transposed, params.out_h, params.out_w, str(dev),
           bool(wino_ok), bool(implicit_ok and IMPLICIT)) + (('split', split_mode()) if split_ok else ()) + (('f43',) if (f43 is not None and wino_ok) else ())
    f43_ok = f43 is not None and wino_ok and not DETERMINISTIC and params.y and not params.y_pool and not params.stats
    f43_t = []

    def f43_w():
        if not f43_t:
            f43_t.append(f43() if callable(f43) else f43)
        return f43_t[0]
    if peek:
        # the answer a later call with the same problem will get, when it is already known (pinned, or in the table): (algo, tile) or None.  Nothing is
        # measured, nothing is written to params
        if FORCE_GRAD is not None and f43 is not None:
            return (6, 5) if (FORCE_GRAD == 'f43' and f43_ok) else (0, 0)
        if FORCE_ALGO is not None or DETERMINISTIC:
            return None
        if str(dev) not in _DEFAULTS_SEEN:
            load_tune_defaults(dev)
        hit = _TUNE.get(key)
        return None if hit is None else (tuple(hit) if isinstance(hit, (list, tuple)) else (0, hit))
    implicit_ok = bool(implicit_ok and IMPLICIT) and params.Cin % 32 == 0
    w_direct = params.w

    def apply(choice):
        algo, tile = choice
        if algo in (1, 2, 3) and wino_w is None:
            raise OperandMissing('algorithm %d reads the Winograd filter transform, which the caller did not prepare' % algo)
        params.algo, params.tile = algo, tile
        params.w = wino_split.data_ptr() if algo in (4, 5) else wino_w.data_ptr() if algo in (1, 2, 3) else f43_w().data_ptr() if algo == 6 else w_direct
        params.w_plane = split_plane if algo in (4, 5) else 0
        # the 4x4-tile filter operand exists only here: it must outlive this function (the caller launches with params AFTER it returns, and a
        # workspace that grows in between would be handed the freed block): the tensor rides on the params object
        params._f43_operand = f43_t if algo == 6 else None
        return choice
    if FORCE_GRAD is not None and f43 is not None:
        # a data gradient under a pinned gradient algorithm (no measurement, no cache)
        if FORCE_GRAD == 'f43' and f43_ok:
            apply((6, 5))
            if lib().y2_conv_fwd_workspace_bytes(ctypes.byref(params)) >= 0:
                return (6, 5)
        return apply((0, 0))
    if FORCE_ALGO is not None:
        # deterministic algorithm coverage (tests, A/B runs): every eligible layer takes the named algorithm, everything else the
        # direct kernel with the library's own tile choice; no measurement, no cache
        want = {'direct': None, 'winograd': (1, 5), 'fused': (2, 0), 'implicit': (3, 0), 'fused3': (2, 3), 'implicit3': (3, 3), 'split': (split_algo(), 0)}[FORCE_ALGO]
        if want is not None and want[0] == 3 and not implicit_ok:
            want = (2, want[1])         # where the transformed input must stay behind: the fused kernel that reads it
        if want is not None and want[0] in (4, 5) and not split_ok:
            want = (1, 5) if wino_ok else None
        if want is not None and wino_ok and (want[0] == 1 or params.Cin % 32 == 0):
            apply(want)
            if lib().y2_conv_fwd_workspace_bytes(ctypes.byref(params)) >= 0:
                return want
        return apply((0, 0))
    if str(dev) not in _DEFAULTS_SEEN:
        load_tune_defaults(dev)
    hit = None if DETERMINISTIC else _TUNE.get(key)      # deterministic mode ignores measured choices (they may differ between runs)
    if hit is not None:
        hit = tuple(hit) if isinstance(hit, (list, tuple)) else (0, hit)
        if PERSIST or not (hit[0] == 0 and hit[1] in (11, 12, 13, 15)):
            return apply(hit)
        # Y2_CONV_PERSIST=0 (A/B switch): a table entry that names a persistent tile does not count - measure (or fall back) without them
    if not AUTOTUNE or DETERMINISTIC or torch.cuda.is_current_stream_capturing():
        # no measurement possible (or, deterministic mode: a timed choice may differ from run to run and with it the rounding): the choices the measurements converge to on MI355X (profiles/r01_detect_b32_layer_table.txt)
        prefer = []
        if f43_ok and params.Cin >= 128 and params.H * params.W <= 19 * 19:
            prefer.append((6, 5))                  # gradients of the 13x13 (19x19) layers: 4x4 tiles, 64x128 GEMM tiles
        if split_ok and params.H * params.W <= 19 * 19:
            prefer.append((split_algo(), 0))       # opt-in split mode: the 13x13 (19x19 at 608) layers, 25-30 % ahead of the fp32 GEMMs there
        if wino_ok and implicit_ok and (params.H * params.W >= 52 * 52 or (params.H * params.W >= 26 * 26 and params.Cout <= params.Cin)):
            if params.Cin <= 128:
                prefer.append((3, 3))   # ... its two-workgroups-per-CU form where the K loop is short (11-13 % ahead on the 104x104 / 52x52 layers)
            prefer.append((3, 0))       # fused Winograd with the input transform in its loader: the large maps and the data gradients
        if wino_ok and params.Cin % 32 == 0 and params.H * params.W >= 26 * 26:
            prefer.append((2, 0))       # fused Winograd on the 104x104 ... 26x26 layers
        if wino_ok and params.Cin >= 128:
            prefer.append((1, 5))       # three-kernel Winograd, 64x128 GEMM tiles, on the 13x13 layers
        for choice in prefer:
            apply(choice)
            if lib().y2_conv_fwd_workspace_bytes(ctypes.byref(params)) >= 0:      # the library accepts this problem (sizes, alignment)
                return choice
        return apply((0, 0))
    L = lib()
    st = stream()
    TUNE_MISSES.append(key)           # a problem shape neither this process nor the committed table has met: measured now (bench.py reports the count per leg)
    big = params.B * params.H * params.W >= 65536 and params.Cout >= 128 and params.ksize in (1, 3) and params.stride in (0, 1) \
        and params.pad_plus1 in (0, (params.ksize - 1) // 2 + 1) and not params.transposed and not params.residual
    cands = [(0, t) for t in [5, 3, 2, 1] + ([6] if params.Cout <= 32 else []) + ([8, 9] if big else [])]      # 8, 9: 512-thread 256x128 / 128x256 tiles
    if PERSIST and params.ksize == 1 and params.stride in (0, 1) and params.pad_plus1 in (0, 1) and not params.transposed and not params.y_pool and params.Cin % 32 == 0:
        cands += [(0, t) for t in (15, 13, 12, 11)]      # 1x1 layers (short K loops): persistent workgroups, next tile's first slab fetched under the current tile's last
    if wino_ok:
        cands += [(1, t) for t in (5, 3, 2, 1)]
        if params.Cin % 32 == 0:
            cands.append((2, 0))        # fused GEMM + output transform (no product tensor): pays on the 52x52 layers
        if implicit_ok:
            cands.append((3, 0))        # ... with the input transform in its loader (no transformed input in memory either)
        if params.Cin % 32 == 0:
            cands.append((2, 3))        # the same two as 32 x 64 units, two workgroups per CU (wino_fused3_kernel): short K loops, small grids
            if implicit_ok:
                cands.append((3, 3))
        if split_ok:
            cands.append((split_algo(), 0))        # three-kernel Winograd with the GEMMs on the bf16 / fp16 pipe (opt-in precision modes)
        if f43_ok:
            cands += [(6, t) for t in (5, 3, 2, 1)]      # gradients: Winograd F(4x4,3x3), three kernels, 36 GEMMs
    best, best_t = (0, 0), float('inf')
    stats_save = params.stats
    params.stats = None          # timing launches must not accumulate statistics twice
    for choice in cands:
        apply(choice)
        if conv_workspace(params, dev) < 0:
            continue
        if L.y2_conv_fwd(ctypes.byref(params), st) != 0:
            continue
        t = _time_conv(L, params, st)
        if t < best_t:
            best, best_t = choice, t
    params.stats = stats_save
    apply(best)
    _TUNE[key] = list(best)
    _tune_save()
    return best


def export_tune():
    """The measured choices of this process as a picklable list of (key, choice); device names replaced by a placeholder."""
    out = []
    for k, v in _TUNE.items():
        out.append((tuple('@dev' if isinstance(e, str) and e.startswith(('cuda', 'cpu')) else e for e in k), v))
    return out


def import_tune(items, dev, merge=False):
    """Adopt another process's choices (export_tune) for device `dev`: every rank of a data-parallel job then runs the same
    algorithms (same rounding, no straggler that measured a worse plan).  merge: entries only this process has (a problem shape
    the other has not met) stay.  Plans built on the old choices are invalidated - only when an entry actually changed."""
    new = {tuple(str(dev) if e == '@dev' else e for e in k): v for k, v in items}
    if not merge:
        if new == _TUNE:
            return
        _TUNE.clear()
        _TUNE.update(new)
        _TUNE_EPOCH[0] += 1
        return
    changed = False
    for k, v in new.items():
        if k not in _TUNE or _same_choice(_TUNE[k], v) is False:
            _TUNE[k] = v
            changed = True
    if changed:
        _TUNE_EPOCH[0] += 1


def _same_choice(a, b):
    return (list(a) if isinstance(a, (list, tuple)) else a) == (list(b) if isinstance(b, (list, tuple)) else b)


def _tune_save():
    if TUNE_CACHE:
        try:
            import json as _json
            _json.dump({_json.dumps(list(k)): v for k, v in _TUNE.items()}, open(TUNE_CACHE, 'w'))
        except Exception:
            pass


_WGRAD_WS = {}


def wgrad_choice(B, H, W, cin, ldx, cout, ldz, k, has_v, dev):
    """Which kernel conv_wgrad will run for this problem: 0 = y2_conv_wgrad (accumulates into a ZEROED buffer), 1 = y2_wino_wgrad
    (2x2 gradient tiles; overwrites), 2 = its 4x4-tile form (Winograd F(3x3, 4x4): fewer multiply-adds, never reads the forward's
    transformed input; overwrites), None = eligible for all and not measured yet (conv_wgrad will time them and zero the buffer itself)."""
    if not (wino_eligible(cout, cin, k) and not (ldx % 4) and not (ldz % 4)):
        return 0
    if FORCE_WGRAD is not None and not DETERMINISTIC:
        return {'direct': 0, 'wino': 1, 'f34': 2}[FORCE_WGRAD]
    if DETERMINISTIC or not AUTOTUNE:
        if not DETERMINISTIC and WGRAD_F34 and cin >= 128 and H * W <= 19 * 19:
            return 2                    # what the measurements converge to on the 13x13 (19x19) layers
        return 1 if cin >= 128 else 0
    if str(dev) not in _DEFAULTS_SEEN:
        load_tune_defaults(dev)
    return _TUNE.get(('wgrad', B, H, W, cin, ldx, cout, ldz, bool(has_v), str(dev)))


def _wgrad_scratch(dev, need):
    """The Winograd weight gradients' scratch (transformed operands, dU): one buffer per device (or per capture scope), grown by replacement.  The weight
    gradients run on the launch stream of launch_on() while this buffer - like everything - is allocated on torch's current stream: before the old buffer
    is dropped the current stream waits for the launch stream, or the allocator would hand its block to the next main-stream tensor while the previous
    layer's weight gradient is still using it (seen as a garbage gradient of one layer in the first backward of a process, where the scratch grows layer by layer)."""
    ws = _cache(_WGRAD_WS).get(str(dev))
    if ws is None or ws.numel() * 4 < need:
        if ws is not None and _LAUNCH is not None:
            torch.cuda.current_stream().wait_stream(_LAUNCH)
        _retire(ws)
        ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=dev)
        _cache(_WGRAD_WS)[str(dev)] = ws
    return ws


def eligible_wino(cout, cin, k, ldx, ldz):
    return wino_eligible(cout, cin, k) and not (ldx % 4) and not (ldz % 4)


def conv_wgrad(x, dz, B, H, W, cin, ldx, cout, ldz, k, v=None, out=None, zeroed=False, native=None, dz_pre=False):
    """Packed weight gradient dw[cout][k*k][cin] of a stride-1 "same" convolution: y2_conv_wgrad (9 shifted reductions
    over pixels) or, for 3x3 layers where it measures faster, y2_wino_wgrad (16 reductions over 2x2 tiles).  The choice
    is timed once per problem shape and cached.  `v`: the layer's transformed input kept from a Winograd forward (see
    include/yolo2_hip.h, y2_wino_wgrad) - the weight gradient then skips its input transform.  `out`: destination (cout*k*k*cin
    floats) instead of a fresh tensor; `zeroed`: the caller has zero-filled it (one y2_multi launch for all layers of a step).
    `native`: a [cout][cin][3][3] destination; when the (already measured) choice is the Winograd reduction it is written directly in
    that layout and returned INSTEAD of the packed buffer (the caller checks `result is native`)."""
    L, st, dev = lib(), stream(), x.device
    nw = cout * cin * k * k
    choice = wgrad_choice(B, H, W, cin, ldx, cout, ldz, k, v is not None, dev)
    if dz_pre:
        # `dz` is the transformed gradient [36][T][cout] of the 4x4-tile form (y2_bn_act_bwd_wino6): the caller looked the choice up before it built it
        if choice != 2:
            raise RuntimeError('conv_wgrad: a transformed gradient serves the 4x4-tile form only (choice %r)' % (choice,))
        need = L.y2_wino_wgrad_workspace_bytes_ex(B, H, W, cin, cout, 6)
        ws = _wgrad_scratch(dev, need)
        dst = native if native is not None else (out if out is not None else torch.empty(nw, dtype=torch.float32, device=dev))
        check(L.y2_wino_wgrad_ex(ptr(x), ptr(dz), ptr(dst), B, H, W, cin, ldx, cout, cout, None, ptr(ws), ws.numel() * 4, 7 if native is not None else 6, st), 'y2_wino_wgrad_ex')
        return dst
    eligible = wino_eligible(cout, cin, k) and not (ldx % 4) and not (ldz % 4)
    key = ('wgrad', B, H, W, cin, ldx, cout, ldz, v is not None, str(dev))
    # the direct kernel accumulates split partial sums into a zeroed buffer; the Winograd path overwrites (no fill needed)
    if native is not None and choice in (1, 2) and eligible_wino(cout, cin, k, ldx, ldz):
        assert native.numel() == nw and native.is_contiguous()
        need = L.y2_wino_wgrad_workspace_bytes_ex(B, H, W, cin, cout, 1 if choice == 1 else 3)
        ws = _wgrad_scratch(dev, need)
        check(L.y2_wino_wgrad_ex(ptr(x), ptr(dz), ptr(native), B, H, W, cin, ldx, cout, ldz, ptr(v) if choice == 1 else None, ptr(ws), ws.numel() * 4,
                                 1 if choice == 1 else 3, st), 'y2_wino_wgrad_ex')
        return native
    dwp = out if out is not None else torch.empty(nw, dtype=torch.float32, device=dev)
    assert dwp.numel() >= nw and dwp.is_contiguous()
    if choice not in (1, 2) and not zeroed:
        dwp.zero_()

    def direct():
        check(L.y2_conv_wgrad(ptr(x), ptr(dz), ptr(dwp), B, H, W, cin, ldx, cout, ldz, k, st), 'y2_conv_wgrad')
    if not eligible:
        direct()
        return dwp
    ws = None
    if choice != 0:      # (a layer whose measured choice is the direct kernel needs no Winograd scratch: the 208x208 layer alone would size it at 4-9 GB)
        need = L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout) if choice is None else L.y2_wino_wgrad_workspace_bytes_ex(B, H, W, cin, cout, 2 if choice == 2 else 0)
        ws = _wgrad_scratch(dev, need)

    def wino():
        check(L.y2_wino_wgrad(ptr(x), ptr(dz), ptr(dwp), B, H, W, cin, ldx, cout, ldz, ptr(v), ptr(ws), ws.numel() * 4, st), 'y2_wino_wgrad')
    def wino6():
        check(L.y2_wino_wgrad_ex(ptr(x), ptr(dz), ptr(dwp), B, H, W, cin, ldx, cout, ldz, None, ptr(ws), ws.numel() * 4, 2, st), 'y2_wino_wgrad_ex')
    if choice is None:
        if torch.cuda.is_current_stream_capturing():
            choice = 1 if cin >= 128 else 0
        else:
            times = []
            for fn in (direct, wino) + ((wino6,) if WGRAD_F34 else ()):
                with _timing_stream():
                    fn()
                    t = float('inf')
                    for _ in range(2):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        fn()
                        fn()
                        e1.record()
                        e1.synchronize()
                        t = min(t, e0.elapsed_time(e1))
                times.append(t)
            if v is not None and H * W >= 52 * 52 and len(times) > 1:
                # The 2x2-tile reduction reads the forward's transformed input V: choosing it makes the FORWARD of this layer keep V, i.e. run
                # the input-transform kernel (|x| read + 4|x| written at ~5.5 TB/s) in front of a V-reading algorithm instead of the implicit
                # one the large maps otherwise take.  That cost belongs to this choice (round 5: on the 208x208 layer the reduction measured
                # 0.93 ms against 0.98 ms for the direct kernel and made the forward 0.32 ms slower - the per-kernel comparison cannot see it;
                # on the 104x104 layers it wins even so).  times are for two launches.
                times[1] += 2.0 * 5.0 * B * H * W * cin * 4 / 5.5e9
            choice = times.index(min(times))
            TUNE_MISSES.append(key)
            _TUNE[key] = choice
            _tune_save()
            with _timing_stream():
                dwp.zero_()          # the timing launches of the direct kernel accumulated into dwp
    (direct, wino, wino6)[choice]()
    return dwp


def f32c(t):
    """Contiguous fp32 view/copy of a GPU tensor."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
