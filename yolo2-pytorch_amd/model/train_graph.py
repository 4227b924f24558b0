"""Training graph of the YOLOv2 hot path on MI355X: autograd.Functions whose forward/backward are chains of HIP kernels.

The reference trains through torch autograd over nn.Conv2d / nn.BatchNorm2d / LeakyReLU / MaxPool2d
(model/yolo2.py:49-130) and elementwise tensor expressions (model/__init__.py:117-167); `train.py:344-357` calls
`_inference` -> `loss` -> `backward` -> `optimizer.step`.  Here the same Python surface is kept and three Functions
carry the arithmetic:

  DarknetTrainFn : x, parameters -> head image (NHWC).  forward = per block {raw conv (+ per-channel sum / sum^2 in the
                   epilogue) -> y2_bn_finalize (batch statistics, running-stat update) -> y2_bn_act_fwd (affine +
                   LeakyReLU + pool / reorg / concat addressing)}; backward = per block, in reverse, {y2_bn_act_bwd
                   (pool routing + LeakyReLU + BN backward, d gamma / d beta) -> y2_conv_wgrad -> dgrad (= y2_conv_fwd on
                   180-degree-rotated, in/out-swapped weights)}.
  DecodeFn       : head image -> iou, center_offset, size_norm, yx_min, yx_max, logits (y2_decode / y2_decode_bwd).
  RegionLossFn   : predictions + labels -> the five loss terms (y2_region_loss_fwd / _bwd).

torch only allocates, keeps the autograd tape and (in train.py's wrapper) runs the RCCL all-reduce.
"""
import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist

import _hip

BWD_STREAMS = int(os.environ.get('Y2_BWD_STREAMS', '2'))     # 2: weight gradients on a side stream (overlap with the HBM-bound passes); 1: one stream
_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(str(dev))
    if s is None:
        s = torch.cuda.Stream(device=dev)
        _SIDE[str(dev)] = s
    return s


# A captured StepPlan may fork the weight gradients onto the side stream like the eager backward does (joined before every segment end).
# Measured at batch 64: the step gains 0.75 ms of GPU time (31.8 -> 31.05 ms at 416x416, 19.8 -> 18.9 at 320x320).  The price is on the host:
# the runtime issues a non-linear graph node by node and hipGraphLaunch returns only when most of the graph has EXECUTED (tools/probes/
# graph_fork_cost.py: 270 small kernels, host time of a launch 0.16 ms linear, 0.84 ms = the graph's GPU time with one fork or with 22;
# 11 ms of a 31 ms training step) - the issuing thread is blocked, not busy.  Y2_GRAPH_FORK: '1' fork, '0' never, 'auto' (default): fork in
# single-process training (the host thread has nothing else to issue during a step), linear under the data-parallel wrapper, where the
# same thread issues the RCCL collectives between graph segments and eight ranks share one host - unmeasured on 8-GPU hardware.
GRAPH_FORK = {'1': True, '0': False}.get(os.environ.get('Y2_GRAPH_FORK', 'auto'), 'auto')
GRAD_F43 = os.environ.get('Y2_GRAD_F43', '1') != '0'        # offer Winograd F(4x4,3x3) to the data gradients of the deep layers (A/B)
FUSE_WINO6 = os.environ.get('Y2_FUSE_WINO6', '1') != '0'      # 0: BatchNorm-backward pass 2 always writes dz and the 4x4-tile transforms read it (A/B; csrc/train.hip: bn_bwd_wino6_kernel)
FUSE_CONV0 = os.environ.get('Y2_FUSE_CONV0', '1') != '0'      # 0: materialise the first layer's dz and run the two-kernel form (A/B runs)
DEBUG_TAP = None        # debugging hook: callable(block name, dz, dx) invoked per block of the Darknet backward

SYNC_POSITIVES = True   # data parallel: all-reduce the positive count so the cls mean is over the global batch

# train.DataParallelRCCL tags the tensors its forward returns with `_y2_dp_reduce`: a callable(tensor) summing a small tensor over
# THAT wrapper's process group, in place (it knows the group and how to reach it: RCCL directly on device memory, or host
# staging under the gloo test backend).  model.loss reads the tag off pred['feature'], so the reduction follows the model the
# predictions came from: an unwrapped model in the same process reduces nothing, two wrappers on two groups do not mix.
DP_TAG = '_y2_dp_reduce'


def _sum_over_ranks(t, reducer):
    """True when `t` was summed over more than one rank."""
    return bool(reducer(t)) if reducer is not None else False


BN_EPS = 1e-5
BN_MOMENTUM = 0.01
LEAKY = 0.1
_WINO_CHUNK_BYTES = int(os.environ.get('Y2_WINO_CHUNK_MB', '4096')) << 20      # csrc/wino.hip: wino_chunk_bytes()


def _counter(bn):
    """nn.BatchNorm2d.num_batches_tracked as y2_bn_finalize increments it in place: an int64 scalar on the module's device."""
    t = bn.num_batches_tracked
    if t is None:
        return None
    if t.dtype != torch.int64 or not t.is_cuda:
        raise RuntimeError('BatchNorm2d.num_batches_tracked must be an int64 GPU tensor (got %s on %s)' % (t.dtype, t.device))
    return t


def _new(dev, *shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=dev)


OperandPruned = _hip.OperandMissing      # a captured step prepares only the GEMM operands its warm-up passes used (StepPlan.used_last); the algorithm table asked for another one


def _conv(L, st, x, wp, y, B, H, W, cin, ldx, cout, k, ldy, scale=None, shift=None, slope=1.0, stats=None, y_pool=None, ldp=0, coff=0, out_mode=0, keep_v=False, u=False,
          us=None, us_plane=0, grad=False, note=None, u_eligible=None, u6=None, peek=False, pre=None, dev=None):
    """One y2_conv_fwd.  keep_v: when the Winograd algorithm is chosen, run it in a workspace of its own and return that tensor -
    its head is the transformed input V, which the weight gradient of the same layer reuses (y2_wino_wgrad v_transformed).
    u: the layer's Winograd filter transform when the caller prepared it (y2_prep_weights), None = not eligible, False = derive it here.
    wp may be None (an operand a captured step did not prepare): choosing an algorithm that reads it raises OperandPruned; likewise
    u = None with u_eligible = True (the layer is Winograd-eligible, its transform was not prepared: the problem keeps its identity in the
    algorithm table - same key, same offer of the 4x4-tile gradient form - and only a choice that READS u fails).
    note: callable('w' | 'u' | '6') told which filter operand the chosen algorithm reads (packed, 2x2-tile transform, 4x4-tile transform).
    u6: the 4x4-tile transform when the caller prepared it; else it is derived from wp when that algorithm is timed or chosen."""
    # peek: return the (algo, tile) this problem is known to take, or None - nothing is launched (x / y may be None).
    # pre: the transformed input [36][T][cin] of the 4x4-tile form, built by y2_bn_act_bwd_wino6 because peek said the problem takes that form: the launch reads
    # it instead of transforming x (which may be None then)
    p = _hip.ConvParams()
    p.x, p.w = (x.data_ptr() if x is not None else 256), (wp.data_ptr() if wp is not None else None)
    p.scale = scale.data_ptr() if scale is not None else None
    p.shift = shift.data_ptr() if shift is not None else None
    p.y = y.data_ptr() if y is not None else (256 if peek else None)
    p.y_pool = y_pool.data_ptr() if y_pool is not None else None
    p.stats = stats.data_ptr() if stats is not None else None
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, H, W, cin, ldx, cout, k
    p.ldy, p.coff, p.ldp, p.poff, p.out_mode = ldy, coff, ldp, 0, out_mode
    p.slope, p.tile = slope, 0
    # 3x3 layers: also offer the Winograd algorithm (filter transform of the packed weight: fprop and dgrad alike)
    if u is False:
        u = _hip.wino_weight(wp, cout, cin) if (out_mode == 0 and _hip.wino_eligible(cout, cin, k)) else None
    elif out_mode != 0:
        u = None
    # us: bf16 plane triple of u (opt-in split-bf16 mode; planes us_plane elements apart), offered as Y2_ALGO_WINOGRAD_SPLIT
    # grad: the output is a data gradient - the deep layers may take the 4x4-tile Winograd form (its filter operand is built on demand)
    def f43_operand():
        if u6 is not None:
            return u6          # prepared with the step's other operands (y2_prep_weights, Y2_PREP_WINO6_DGRAD)
        if wp is None:
            raise OperandPruned('the 4x4-tile data-gradient operand is derived from a packed weight this step did not prepare')
        return _hip.wino6_weight(wp, cout, cin)
    eligible = (u is not None) if u_eligible is None else (bool(u_eligible) and out_mode == 0)
    f43 = f43_operand if (grad and GRAD_F43 and eligible and cin >= 128 and H * W <= 52 * 52) else None      # (offered; the measurement decides: 13x13 ... 26x26 at 416, 19x19 ... 38x38 at 608)
    dev = dev if dev is not None else (x if x is not None else (pre if pre is not None else y)).device
    if peek:
        return _hip.autotune_conv(p, dev, wino_w=u, implicit_ok=not keep_v, wino_split=us if u is not None else None, split_plane=us_plane, f43=f43, wino_eligible=eligible, peek=True)
    _hip.autotune_conv(p, dev, wino_w=u, implicit_ok=not keep_v, wino_split=us if u is not None else None, split_plane=us_plane, f43=f43, wino_eligible=eligible)
    if pre is not None:
        if p.algo != 6:
            raise RuntimeError('_conv: a transformed input serves the 4x4-tile form only (algorithm %d chosen)' % p.algo)
        p.algo, p.x, p.ldx = 7, pre.data_ptr(), cin          # Y2_ALGO_WINOGRAD_F43_PRE
    if wp is None and (p.algo == 0 or (p.algo in (6, 7) and u6 is None)):
        raise OperandPruned('algorithm %d reads the packed weight, which this step did not prepare' % p.algo)
    if note is not None:
        note('w' if p.algo == 0 else '6' if p.algo in (6, 7) else 'u')
    kept = None
    if keep_v and p.algo in (1, 2):
        T = B * ((H + 1) // 2) * ((W + 1) // 2)
        if 16 * T * (cin + cout) * 4 <= _WINO_CHUNK_BYTES and 16 * T * cin * 4 < 0x7fffffff:        # one batch chunk: V covers the whole batch
            need = L.y2_conv_fwd_workspace_bytes(ctypes.byref(p))
            kept = torch.empty(need // 4 + 4, dtype=torch.float32, device=x.device)
            p.workspace, p.workspace_bytes = kept.data_ptr(), kept.numel() * 4
    if kept is None:
        _hip.conv_workspace(p, dev)
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), st), 'y2_conv_fwd')
    return kept


class _Block(object):
    """One conv block of the forward pass: geometry + saved tensors for backward."""
    __slots__ = ('mod', 'name', 'x', 'ldx', 'H', 'W', 'cin', 'cout', 'k', 'z', 'scale', 'shift', 'mean', 'invstd', 'pool',
                 'out_full', 'out_pool', 'out_ld', 'out_off', 'out_mode', 'has_bn', 'slope', 'first', 'wino_v', 'eff')


PRUNE_OPERANDS = os.environ.get('Y2_PRUNE_OPERANDS', '1') != '0'      # 0: a captured step derives all four operand forms of every layer (A/B)


def _train_operands(dnn, dev, scope=None, only=None, alloc_only=False):
    """GEMM operands of every convolution block for one parameter version, produced by ONE y2_prep_weights launch from the
    state_dict layout: fprop pack, dgrad pack (rotated, in/out swapped) and, where the Winograd algorithm is eligible, both filter
    transforms.  The optimizer rewrites the weights every step, so this runs once per step (it used to be ~90 separate small
    launches: pack + transform per layer, forward and backward).  Returns {Conv2d block: dict(wp, wd, uf, ud)}; the first layer
    (y2_conv0_fwd reads the state_dict layout) and blocks whose output width is not a multiple of 4 (the 125 / 425-channel head: its
    data gradient runs zero-padded) are left to the per-layer path.
    scope: the buffer dict of a StepPlan - the operands are derived unconditionally (the launch must be part of every replay of the plan's
    graph) into buffers that plan owns; the per-model cache is neither read nor written.
    only: set of (block, 'wp' | 'wd' | 'uf' | 'ud' | 'u6d') - derive just these ('u6d': the 4x4-tile data-gradient operand, prepared here
    only on request: the eager passes derive it on demand from 'wd') (a StepPlan knows from its warm-up passes which operand each
    layer's chosen algorithms read: typically ONE of the two forward forms and ONE of the two data-gradient forms, i.e. half of the
    2.2 GB this launch moves per step); the others stay None.  alloc_only: make sure the buffers exist, launch nothing (a plan allocates
    shared operand buffers before its capture begins, outside the graph's memory pool)."""
    from model import yolo2 as _yolo2
    if _hip.split_mode() or scope is None:
        only = None          # (the split modes derive their planes from the whole arena; the per-model cache serves the autograd path with everything)
    key = (dev, dnn._weight_versions(), _hip.split_mode(), _hip.WINOGRAD)       # the convolution weights only: the BatchNorm buffer updates of a forward pass do not move it
    if scope is not None:
        bufs = (dev, scope)
    else:
        cache = getattr(dnn, '_train_cache', None)
        if cache is not None and cache[0] == key:
            return cache[1]
        bufs = getattr(dnn, '_train_bufs', None)
        if bufs is None or bufs[0] != dev:
            bufs = (dev, {})
            dnn._train_bufs = bufs

    def buf(tag, n):
        t = bufs[1].get(tag)
        if t is None or t.numel() != n or t.device != dev:
            t = torch.empty(n, dtype=torch.float32, device=dev)
            bufs[1][tag] = t
        return t
    first = dnn._first_block()
    items, ops = [], {}
    # the Winograd filter transforms of all layers live in ONE arena: in the split-bf16 mode a single y2_split_bf16x3 pass turns it into
    # the three bf16 planes of every layer's operand (planes `usize` elements apart)
    wino = []
    for name, blk in dnn.named_modules():
        if not isinstance(blk, _yolo2.Conv2d) or blk is first:
            continue
        w = blk.conv.weight.detach()
        cout, cin, k, _ = w.shape
        if cout % 4 or cin % 4 or not w.is_contiguous() or w.dtype != torch.float32:
            continue
        n = w.numel()
        want = lambda tag: only is None or (blk, tag) in only
        d = dict(name=name, wp=buf((name, 'wp'), n) if want('wp') else None, wd=buf((name, 'wd'), n) if want('wd') else None, uf=None, ud=None, ufs=None, uds=None, plane=0,
                 uf_ok=bool(_hip.wino_eligible(cout, cin, k)), ud_ok=bool(_hip.wino_eligible(cin, cout, k)))      # eligibility does not depend on what `only` leaves out
        if d['wp'] is not None:
            items.append((w, d['wp'], cout, cin, k, _hip.PREP_FPROP))
        if d['wd'] is not None:
            items.append((w, d['wd'], cout, cin, k, _hip.PREP_DGRAD))
        if _hip.wino_eligible(cout, cin, k) and want('uf'):
            wino.append((d, 'uf', w, cout, cin, k, _hip.PREP_WINO_FPROP))
        if _hip.wino_eligible(cin, cout, k) and want('ud'):          # the data gradient is a convolution with the roles of Cin and Cout exchanged
            wino.append((d, 'ud', w, cout, cin, k, _hip.PREP_WINO_DGRAD))
        d['u6d'] = None
        if only is not None and (blk, 'u6d') in only and d['ud_ok']:
            d['u6d'] = buf((name, 'u6d'), 36 * cout * cin)
            items.append((w, d['u6d'], cout, cin, k, _hip.PREP_WINO6_DGRAD))
        ops[blk] = d
    usize = sum(16 * cout * cin for _, _, _, cout, cin, _, _ in wino)
    if usize:
        if only is None:
            arena = buf('u_arena', usize)
        off = 0
        for d, tag, w, cout, cin, k, mode in wino:
            # (a pruned set: one buffer per operand instead of the arena - plans of different input sizes share them by name, and the
            # split modes, which turn the whole arena into planes with one pass, never prune)
            d[tag] = arena[off:off + 16 * cout * cin] if only is None else buf((d['name'], tag), 16 * cout * cin)
            items.append((w, d[tag], cout, cin, k, mode))
            off += 16 * cout * cin
    if alloc_only:
        return ops
    if items:
        table = (_hip.PrepItem * len(items))()
        for e, (src, dst, cout, cin, k, mode) in zip(table, items):
            e.src, e.dst, e.Cout, e.Cin, e.ksize, e.mode = src.data_ptr(), dst.data_ptr(), cout, cin, k, mode
        _hip.check(_hip.lib().y2_prep_weights(table, len(items), _hip.stream()), 'y2_prep_weights')
    if usize and _hip.split_mode():
        f16 = _hip.split_mode() == 'f16'
        np_, dt = (2, torch.float16) if f16 else (3, torch.bfloat16)
        planes = bufs[1].get('u_split')
        if planes is None or planes.numel() != np_ * usize or planes.dtype != dt:
            planes = bufs[1]['u_split'] = torch.empty(np_ * usize, dtype=dt, device=dev)
        if f16:
            _hip.check(_hip.lib().y2_split_f16x2(_hip.ptr(arena), _hip.ptr(planes), usize, _hip.F16_U_SCALE, _hip.stream()), 'y2_split_f16x2')
        else:
            _hip.check(_hip.lib().y2_split_bf16x3(_hip.ptr(arena), _hip.ptr(planes), usize, _hip.stream()), 'y2_split_bf16x3')
        off = 0
        for d, tag, w, cout, cin, k, mode in wino:
            if (cout if tag == 'ud' else cin) % 32 == 0:          # the K dimension of the GEMM (Cin of the convolution that runs)
                d[tag + 's'] = planes[off:off + 16 * cout * cin]
            d['plane'] = usize
            off += 16 * cout * cin
    if scope is None:
        dnn._train_cache = (key, ops)
    return ops


def darknet_forward(dnn, x):
    """Training-mode forward of model.yolo2.Darknet; returns the NCHW view like the inference path."""
    params = [p for p in dnn.parameters()]
    out = DarknetTrainFn.apply(dnn, x, *params)
    return out.permute(0, 3, 1, 2)


def darknet_forward_eval_grad(dnn, x):
    """eval()-mode forward that stays differentiable (nn.Module semantics: the reference back-propagates through `dnn` in eval mode,
    receptive_field_analyzer.py:67,87; frozen-BatchNorm fine-tuning).  Forward = the inference chain (folded BatchNorm, nothing
    saved but the input); backward re-runs the network through the training graph with FROZEN statistics and back-propagates."""
    params = [p for p in dnn.parameters()]
    out = DarknetEvalGradFn.apply(dnn, x, *params)
    return out.permute(0, 3, 1, 2)


class _Eff(object):
    """Effective parameters of one conv block for one pass: the module's own tensors, or - for channel counts that are not
    multiples of 4 (pruned checkpoints, model/__init__.py:29-43) - zero-padded copies living in a 4-aligned channel space."""
    __slots__ = ('w', 'gamma', 'beta', 'rm', 'rv', 'bias', 'cout', 'cin', 'cout_r', 'cin_r', 'in_idx', 'padded')


def _pad_layout(dnn):
    """{Conv2d block: (cout_e, cin_e, in_idx)} when some width of the network is not a multiple of 4, else None.  The kernels of the
    training path (LDS-DMA operand loads, vectorised BatchNorm passes) want 4-aligned channel strides, so such a network runs in a
    zero-padded channel space: padded output channels are exactly zero through conv / BatchNorm (gamma 1, beta 0) / LeakyReLU / pool,
    padded input channels meet zero weights, and every gradient of a padded element is exactly zero - the real slices are what
    autograd sees.  in_idx: position of every real input channel in the padded input (None = a plain prefix); the concat buffer
    [4 * c_pt | c_l2] (model/yolo2.py:126-129) interleaves the padding of the reorg'ed passthrough copies."""
    b1, b2, b3 = dnn._blocks()
    up = lambda c: (c + 3) // 4 * 4
    mods = [m for _, m, _ in b1 + b2 + b3] + [dnn.passthrough]
    head = b3[-1][1]
    if all(m.conv.weight.shape[0] % 4 == 0 or m is head for m in mods) and all(m.conv.weight.shape[1] % 4 == 0 for m in mods[1:]):
        return None
    lay = {}
    prev = None
    for i, (_, m, _) in enumerate(b1):
        cout, cin = m.conv.weight.shape[:2]
        lay[m] = (up(cout), cin if i == 0 else prev, None)
        prev = up(cout)
    route = prev
    m = dnn.passthrough
    c_pt = m.conv.weight.shape[0]
    lay[m] = (up(c_pt), route, None)
    for _, m, _ in b2:
        lay[m] = (up(m.conv.weight.shape[0]), prev, None)
        prev = up(m.conv.weight.shape[0])
    c_l2 = b2[-1][1].conv.weight.shape[0]
    idx = [q * up(c_pt) + c for q in range(4) for c in range(c_pt)] + [4 * up(c_pt) + c for c in range(c_l2)]
    m = b3[0][1]
    lay[m] = (up(m.conv.weight.shape[0]), 4 * up(c_pt) + up(c_l2), idx)
    lay[head] = (head.conv.weight.shape[0], up(m.conv.weight.shape[0]), None)       # the head's own width stays (its gradient runs zero-padded, see backward)
    return lay


def _effective(dnn, dev, frozen):
    """{block: _Eff} for one pass (see _pad_layout)."""
    lay = _pad_layout(dnn)
    out = {}
    for m in dnn.modules():
        if not hasattr(m, 'conv') or not hasattr(m, 'has_act'):
            continue
        e = _Eff()
        w = m.conv.weight.detach()
        cout, cin = w.shape[:2]
        e.cout_r, e.cin_r = cout, cin
        e.cout, e.cin, e.in_idx = (cout, cin, None) if lay is None else lay[m]
        e.padded = (e.cout, e.cin) != (cout, cin)
        bn = m.bn
        if not e.padded:
            e.w = _hip.f32c(w)
            e.gamma, e.beta = (bn.weight.detach(), bn.bias.detach()) if bn is not None else (None, None)
            e.rm, e.rv = (bn.running_mean, bn.running_var) if bn is not None else (None, None)
            e.bias = _hip.f32c(m.conv.bias.detach()) if m.conv.bias is not None else None
        else:
            # host-side glue on the pruned-model path only: scatter the real rows / columns into zero tensors of the padded shape
            k = w.shape[2]
            e.w = torch.zeros(e.cout, e.cin, k, k, dtype=torch.float32, device=dev)
            if e.in_idx is None:
                e.w[:cout, :cin] = w
            else:
                e.w[:cout].index_copy_(1, torch.tensor(e.in_idx, dtype=torch.long, device=dev), _hip.f32c(w))
            pad = lambda t, fill: torch.cat([_hip.f32c(t.detach()), torch.full((e.cout - cout,), fill, dtype=torch.float32, device=dev)])
            e.gamma, e.beta = (pad(bn.weight, 1.0), pad(bn.bias, 0.0)) if bn is not None else (None, None)
            e.rm, e.rv = (pad(bn.running_mean, 0.0), pad(bn.running_var, 1.0)) if bn is not None else (None, None)
            e.bias = pad(m.conv.bias, 0.0) if m.conv.bias is not None else None
        out[m] = e
    return out


def _darknet_fwd(ctx, dnn, x, params, frozen, scope=None):
    _hip.require_gpu(x)
    L = _hip.lib()
    st = _hip.stream()
    x = _hip.f32c(x.detach())
    B, cin0, H, W = x.shape
    if H % 32 or W % 32:
        raise ValueError('input size must be a multiple of 32 (got %dx%d)' % (H, W))
    dev = x.device
    b1, b2, b3 = dnn._blocks()
    eff = _effective(dnn, dev, frozen)
    used = ctx.used = set()          # (block, operand form) pairs this pass's chosen algorithms read: what a captured step has to prepare
    prepared = _train_operands(dnn, dev, (getattr(ctx, 'ops_scope', None) or scope) if scope is not None else None, only=getattr(ctx, 'only', None)) if not any(e.padded for e in eff.values()) else {}
    det = _hip.ensure_deterministic(dev)      # fixed-order reductions: BN statistics by y2_colstats_det instead of epilogue atomics
    # one zero-filled arena for every layer's replicated BN-statistics accumulators (one launch instead of 22 fills)
    arena = None
    if not frozen:
        arena = torch.empty(_hip.STATS_REPL * 2 * sum(e.cout for e in eff.values()), dtype=torch.float64, device=dev)
        _hip.multi([(_hip.MULTI_ZERO, arena, None)])
    arena_used = [0]

    def take(n):
        t = arena[arena_used[0]:arena_used[0] + n]
        arena_used[0] += n
        return t
    blocks = []
    written = []          # running statistics / step counters updated through raw pointers by y2_bn_finalize

    def keep_v(h, w, cin, ldx, cout, k):
        """Should the forward convolution leave its transformed input behind?  Only the Winograd weight gradient reads it: once that
        layer's weight gradient is known to run the direct kernel (measured in the first backward with V at hand; the 208x208 layer),
        the forward is free to take the algorithm that never materialises V."""
        if k != 3:
            return False
        with_v, without_v = (_hip.wgrad_choice(B, h, w, cin, ldx, cout, cout, k, hv, dev) for hv in (True, False))
        # (0 = direct kernel, 2 = the 4x4-tile Winograd form: neither reads V; never had V and one of them won anyway: stays that way)
        return not (with_v in (0, 2) or (with_v is None and without_v in (0, 2)))

    def run_block(name, mod, xin, ldx, h, w, pool, out_full=None, out_ld=0, out_off=0, out_mode=0, want_full=True, first=False):
        """raw conv + stats -> finalize -> act.  Returns (_Block, full activation or None, pooled activation or None)."""
        blk = _Block()
        e = eff[mod]
        L.y2_prof_set_tag(1 + len(blocks))          # measurement hooks: forward launches of block i carry tag 1 + i
        cout, cin, k = e.cout, e.cin, mod.kernel_size
        blk.mod, blk.name, blk.x, blk.ldx, blk.H, blk.W, blk.cin, blk.cout, blk.k = mod, name, xin, ldx, h, w, cin, cout, k
        blk.pool, blk.has_bn, blk.slope, blk.first = pool, mod.bn is not None, (LEAKY if mod.has_act else 1.0), first
        blk.wino_v = None
        blk.eff = e
        z = _new(dev, B, h, w, cout)
        stats = take(_hip.STATS_REPL * 2 * cout) if (blk.has_bn and not frozen) else None
        estats = None if det else stats          # statistics accumulated by the convolution's epilogue (atomics)
        if first:
            _hip.check(L.y2_conv0_fwd(_hip.ptr(xin), _hip.ptr(e.w), None, None, _hip.ptr(z), None, _hip.ptr(estats),
                                      B, h, w, cin, cout, cout, 0, 1.0, st), 'y2_conv0_fwd')
        elif mod in prepared:
            blk.wino_v = _conv(L, st, xin, prepared[mod]['wp'], z, B, h, w, cin, ldx, cout, k, cout, stats=estats, keep_v=keep_v(h, w, cin, ldx, cout, k), u=prepared[mod]['uf'],
                               us=prepared[mod]['ufs'], us_plane=prepared[mod]['plane'], note=lambda kind, m=mod: used.add((m, 'uf' if kind == 'u' else 'wp')), u_eligible=prepared[mod]['uf_ok'])
        else:
            wp = _new(dev, e.w.numel())
            _hip.check(L.y2_pack_weight(_hip.ptr(e.w), _hip.ptr(wp), cout, cin, k, 0, st), 'y2_pack_weight')
            blk.wino_v = _conv(L, st, xin, wp, z, B, h, w, cin, ldx, cout, k, cout, stats=estats, keep_v=keep_v(h, w, cin, ldx, cout, k))
        if det and stats is not None:
            _hip.colstats_det(z, B * h * w, cout, cout, stats)
        blk.z = z
        if blk.has_bn and frozen:
            # eval-mode BatchNorm: the folded affine of the inference path; z-hat is built from the running statistics
            blk.scale, blk.shift = _new(dev, cout), _new(dev, cout)
            _hip.check(L.y2_bn_fold(_hip.ptr(e.gamma), _hip.ptr(e.beta), _hip.ptr(e.rm), _hip.ptr(e.rv), BN_EPS, _hip.ptr(blk.scale), _hip.ptr(blk.shift), cout, st), 'y2_bn_fold')
            blk.mean, blk.invstd = _hip.f32c(e.rm), torch.rsqrt(_hip.f32c(e.rv) + BN_EPS)
        elif blk.has_bn:
            bn = mod.bn
            blk.scale, blk.shift, blk.mean, blk.invstd = (_new(dev, cout) for _ in range(4))
            _hip.check(L.y2_bn_finalize(_hip.ptr(stats), float(B * h * w), _hip.ptr(e.gamma), _hip.ptr(e.beta),
                                        _hip.ptr(e.rm), _hip.ptr(e.rv), BN_MOMENTUM, BN_EPS,
                                        _hip.ptr(blk.scale), _hip.ptr(blk.shift), _hip.ptr(blk.mean), _hip.ptr(blk.invstd), cout,
                                        _hip.ptr(_counter(bn)), st), 'y2_bn_finalize')
            if e.padded and bn.running_mean is not None:
                bn.running_mean.copy_(e.rm[:e.cout_r])
                bn.running_var.copy_(e.rv[:e.cout_r])
                written.extend(t for t in (bn.num_batches_tracked,) if t is not None)
            else:
                written.extend(t for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked) if t is not None)
        else:
            blk.scale, blk.mean, blk.invstd = None, None, None
            blk.shift = e.bias
        y_full = y_pool = None
        if out_full is not None:
            y_full = out_full
        elif want_full:
            y_full, out_ld, out_off = _new(dev, B, h, w, cout), cout, 0
        if pool:
            y_pool = _new(dev, B, h // 2, w // 2, cout)
        _hip.check(L.y2_bn_act_fwd(_hip.ptr(z), _hip.ptr(blk.scale), _hip.ptr(blk.shift), blk.slope, _hip.ptr(y_full), _hip.ptr(y_pool),
                                   B, h, w, cout, cout, out_ld, out_off, cout, 0, out_mode, st), 'y2_bn_act_fwd')
        blk.out_full, blk.out_pool, blk.out_ld, blk.out_off, blk.out_mode = y_full, y_pool, out_ld, out_off, out_mode
        blocks.append(blk)
        return blk, y_full, y_pool

    # ---- layers1 (model/yolo2.py:76-96)
    cur, ld, h, w = x, cin0, H, W
    full_last = None
    for i, (name, mod, pool) in enumerate(b1):
        last = i == len(b1) - 1
        pool = pool or last      # layers2 starts with the MaxPool that follows layers1[-1] (model/yolo2.py:97)
        blk, yf, yp = run_block(name, mod, cur, ld, h, w, pool, want_full=(not pool) or last, first=(i == 0))
        if last:
            full_last, fh, fw = yf, h, w
        if pool:
            cur, h, w = yp, h // 2, w // 2
        else:
            cur = yf
        ld = blk.cout
    # ---- passthrough + reorg into the concat buffer (model/yolo2.py:107,126,129)
    c_pt = eff[dnn.passthrough].cout
    c_l2 = eff[b2[-1][1]].cout
    cat = _new(dev, B, h, w, 4 * c_pt + c_l2)
    run_block('passthrough', dnn.passthrough, full_last, full_last.shape[-1], fh, fw, False, out_full=cat, out_ld=cat.shape[-1], out_off=0, out_mode=1)
    # ---- layers2 (leading MaxPool already applied: `cur` is the pooled output of layers1[-1])
    for i, (name, mod, pool) in enumerate(b2):
        if i == len(b2) - 1:
            blk, yf, yp = run_block(name, mod, cur, ld, h, w, False, out_full=cat, out_ld=cat.shape[-1], out_off=4 * c_pt)
        else:
            blk, yf, yp = run_block(name, mod, cur, ld, h, w, False)
            cur, ld = yf, blk.cout
    # ---- layers3
    cur, ld = cat, cat.shape[-1]
    for i, (name, mod, pool) in enumerate(b3):
        blk, yf, yp = run_block(name, mod, cur, ld, h, w, False)
        cur, ld = yf, blk.cout
    if written:
        _hip.wrote(written)
    ctx.dnn = dnn
    ctx.blocks = blocks
    ctx.prepared = prepared
    ctx.prepared_key = dnn._train_cache[0] if (prepared and scope is None) else None
    ctx.scope = scope
    ctx.geom = (B, cin0, H, W, c_pt, c_l2)
    ctx.x = x
    ctx.frozen = frozen
    ctx.param_ids = [id(p) for p in params]
    L.y2_prof_set_tag(0)
    return cur


def _darknet_bwd(ctx, dout):
    L = _hip.lib()
    st = _hip.stream()
    dnn, blocks = ctx.dnn, ctx.blocks
    scope = getattr(ctx, 'scope', None)       # a StepPlan's pass: its operands were derived inside this very launch sequence, into buffers the plan owns
    if ctx.prepared and scope is None and (getattr(dnn, '_train_cache', (None, None))[0] != ctx.prepared_key or ctx.prepared_key[:2] != (dout.device, dnn._weight_versions())):
        raise RuntimeError('model.yolo2: a convolution weight was modified (optimizer step, load_state_dict, in-place edit) between this forward and '
                           'its backward; the per-model GEMM operand buffers this graph was recorded against hold other weights now')
    B, cin0, H, W, c_pt, c_l2 = ctx.geom
    dev = dout.device
    dout = _hip.f32c(dout)
    grads = {}
    hook = getattr(dnn, 'grad_ready_hook', None)
    buffer_hook = getattr(dnn, 'grad_buffer_hook', None)     # train.DataParallelRCCL: where the averaged gradient of a parameter will live (its flat-bucket slice)

    def ready(param, g):
        grads[id(param)] = g
        if hook is not None:
            hook(param, g)

    def dest(param):
        """Tensor a finished gradient of `param` is written to: the data-parallel bucket slice when the wrapper offers one (the
        all-reduce then runs in place, no copy into the bucket), else fresh memory."""
        t = buffer_hook(param) if buffer_hook is not None else None
        return t if t is not None else _new(dev, *param.shape)

    # gradient sources per block index: (dy_full tensor, ldf, foff, fmode), dy_pool tensor
    n = len(blocks)
    src_full = [None] * n
    src_pool = [None] * n
    idx = {b.name: i for i, b in enumerate(blocks)}
    head = n - 1
    src_full[head] = (dout, blocks[head].cout, 0, 0)
    n1 = len(dnn._blocks()[0])
    i_pass = idx['passthrough']
    order = list(range(n - 1, -1, -1))
    main = torch.cuda.current_stream(dev)
    # (under capture only a StepPlan - which joins the side stream before it ends a graph segment, ctx.join - may fork: an unjoined stream fails the capture)
    side = _side_stream(dev) if (BWD_STREAMS > 1 and not _hip.DETERMINISTIC and (not torch.cuda.is_current_stream_capturing() or getattr(ctx, 'fork_ok', False))) else None
    late = []                                         # (parameter, gradient, event) of weight gradients still running on the side stream

    def join():
        for item in late:
            if item[2] is not None:
                main.wait_event(item[2])
                item[2] = None           # (an event recorded in a capture that has ended must not be waited for in the next one)
    ctx.join = join
    affine_grads = []                                 # (parameter, offset into sums_arena, length)

    # ---- everything that must start from zero, filled by ONE launch: the fp64 sums of all BatchNorm backward passes, the
    # accumulation targets of the direct (split, atomically added) weight gradients and the zero-padded gradient of an unaligned head
    sums_arena = torch.empty(2 * sum(b.cout for b in blocks), dtype=torch.float64, device=dev)
    zero = [sums_arena]
    if scope is not None:
        bufs = (dev, scope)
    else:
        bufs = dnn.__dict__.setdefault('_train_bufs', (dev, {}))
        if bufs[0] != dev:
            bufs = dnn._train_bufs = (dev, {})

    def persistent(tag, nel):
        t = bufs[1].get(tag)
        if t is None or t.numel() != nel:
            t = bufs[1][tag] = torch.empty(nel, dtype=torch.float32, device=dev)
        return t
    wg = {}          # block index -> (accumulation target, it is the final gradient tensor, pre-zeroed)
    dzs = {}
    for i in order:
        blk = blocks[i]
        e = blk.eff
        cop = (blk.cout + 3) // 4 * 4
        weight = blk.mod.conv.weight
        if cop != blk.cout:
            dzs[i] = _new(dev, B, blk.H, blk.W, cop)
            zero.append(dzs[i])
        if blk.first:
            if blk.cin <= 3 and blk.cout <= 64:
                t = dest(weight) if not e.padded else _new(dev, blk.cout, blk.cin, 3, 3)
                wg[i] = (t, True, True)
                zero.append(t)
            continue
        choice = _hip.wgrad_choice(B, blk.H, blk.W, blk.cin, blk.ldx, cop, cop, blk.k, blk.wino_v is not None, dev)
        final = blk.k == 1 and cop == blk.cout and not e.padded        # [cout][1][cin] IS the state_dict layout: no unpack pass
        if blk.k == 1:      # (never a persistent buffer: what this kernel writes is handed to autograd as it is, or as a prefix view)
            t = dest(weight).view(-1) if final else _new(dev, cop * blk.cin)
        else:               # packed [cout][tap][cin] staging of a 3x3 gradient, unpacked into the gradient tensor afterwards: reused every step
            t = persistent((blk.name, 'dwp'), cop * blk.cin * blk.k * blk.k)
        if choice == 0:
            zero.append(t)
        wg[i] = (t, final, choice == 0, choice)
    _hip.multi([(_hip.MULTI_ZERO, t, None) for t in zero], st)
    sums_used = 0

    def flush_weight_grads(keep=0):
        while len(late) > keep:
            prm, g, evt, held = late.pop(0)
            if evt is not None:
                main.wait_event(evt)
            del held                 # (the main stream is behind the side stream's reads now: these blocks may be reused by what it launches next)
            ready(prm, g)
    for i in order:
        blk = blocks[i]
        e = blk.eff
        L.y2_prof_set_tag(101 + i)                  # backward launches of block i: tag 101 + i
        h, w, cout, cin, k = blk.H, blk.W, blk.cout, blk.cin, blk.k
        sums = sums_arena[sums_used:sums_used + 2 * cout]
        sums_used += 2 * cout
        # the wgrad / dgrad DMA kernels want channel counts that are multiples of 4: an unaligned Cout (the 125 / 425 channel head)
        # is handled in a zero-padded channel space here; unaligned widths elsewhere were padded by the forward (_pad_layout)
        cop = (cout + 3) // 4 * 4
        sf, sp = src_full[i], src_pool[i]
        # first layer (model/yolo2.py:78-79: conv + pool, nothing below it needs a data gradient): its dz is consumed by the weight gradient
        # alone, which forms it on the fly from (z, dy_pool) and the pass-1 sums - no dz tensor (1.4 GB at batch 64), no second pass
        need_dx = bool(getattr(ctx, 'need_dx', False))
        fuse0 = (FUSE_CONV0 and blk.first and i in wg and sf is None and sp is not None and not (h & 1) and not (w & 15)
                 and B * h * w * cout * 4 < 0xffff0000 and not need_dx)
        # A deep 3x3 block whose weight gradient AND data gradient both run the 4x4-tile Winograd forms reads dz only through their two transforms:
        # y2_bn_act_bwd_wino6 writes those instead of dz (pass 2 + wino6_in x 2 in one kernel, bit-identical operands; csrc/train.hip).  Known in advance
        # from the algorithm table (nothing is measured here: an unknown problem takes the three-kernel form and gets measured there).
        ready_ops = ctx.prepared.get(blk.mod) if not blk.first else None
        fused6 = None
        if (FUSE_WINO6 and k == 3 and not blk.first and ready_ops is not None and sp is None and sf is not None and sf[3] == 0 and cop == cout and not e.padded
                and i in wg and wg[i][3] == 2 and DEBUG_TAP is None and not _hip.DETERMINISTIC and not _hip.split_mode()):
            hit = _conv(L, st, None, ready_ops['wd'], None, B, h, w, cop, cop, cin, k, cin, u=ready_ops['ud'], grad=True, u_eligible=ready_ops['ud_ok'], u6=ready_ops.get('u6d'), peek=True, dev=dev)
            if hit is not None and hit[0] == 6:
                T6 = int(L.y2_wino6_tiles(B, h, w))
                fused6 = (_new(dev, 36 * T6 * cout), _new(dev, 36 * T6 * cout))
        dz = None if (fuse0 or fused6) else (dzs[i] if i in dzs else _new(dev, B, h, w, cop))
        if DEBUG_TAP is not None:
            DEBUG_TAP(blk.name + ':in', blk.z, blk.shift, sf, sp)
        if fused6:
            _hip.check(L.y2_bn_act_bwd_wino6(_hip.ptr(blk.z), _hip.ptr(blk.scale), _hip.ptr(blk.shift), _hip.ptr(blk.mean), _hip.ptr(blk.invstd),
                                             _hip.ptr(e.gamma) if blk.has_bn else None, blk.slope, _hip.ptr(sf[0]), sf[1], sf[2], _hip.ptr(sums),
                                             _hip.ptr(fused6[0]), _hip.ptr(fused6[1]), None, 0, B, h, w, cout, cout,
                                             (2 if ctx.frozen else 1) if blk.has_bn else 0, st), 'y2_bn_act_bwd_wino6')
        else:
            _hip.check(L.y2_bn_act_bwd(_hip.ptr(blk.z), _hip.ptr(blk.scale), _hip.ptr(blk.shift), _hip.ptr(blk.mean), _hip.ptr(blk.invstd),
                                       _hip.ptr(e.gamma) if blk.has_bn else None, blk.slope,
                                       _hip.ptr(sf[0]) if sf else None, sf[1] if sf else 0, sf[2] if sf else 0, sf[3] if sf else 0,
                                       _hip.ptr(sp), cout, 0, _hip.ptr(sums), _hip.ptr(dz), cop, B, h, w, cout, cout,
                                       (2 if ctx.frozen else 1) if blk.has_bn else 0, st), 'y2_bn_act_bwd')
        # parameter gradients of the affine part = the fp64 sums of pass 1: converted for ALL layers by one launch after the loop
        # (they are a few KB per layer; 23 separate 5-microsecond conversions were pure launch latency)
        if blk.has_bn:
            affine_grads.append((blk.mod.bn.bias, sums_used - 2 * cout, e.cout_r))
            affine_grads.append((blk.mod.bn.weight, sums_used - cout, e.cout_r))
        elif blk.mod.conv.bias is not None:
            affine_grads.append((blk.mod.conv.bias, sums_used - 2 * cout, e.cout_r))
        # weight gradient: off the critical path (nothing downstream of this layer's backward needs it), so it runs on a SIDE stream
        # and its MFMA-bound kernel overlaps the HBM-bound passes (y2_bn_act_bwd, Winograd input transforms) of the layers that
        # follow on the main stream.  The gradient is handed to autograd / the data-parallel hook one layer later, behind an event.
        weight = blk.mod.conv.weight

        def real(dw):
            """[cop][cin_e][k][k] in the (padded) channel space of the pass -> the parameter's own shape."""
            if not e.padded and cop == cout:
                return dw
            dw = dw[:e.cout_r]                       # a prefix of the leading dimension: contiguous, no copy
            if e.cin != e.cin_r:
                dw = dw.index_select(1, torch.tensor(e.in_idx, dtype=torch.long, device=dev)) if e.in_idx is not None else dw[:, :e.cin_r].contiguous()
            return dw

        def weight_grad(st_w):
            if i in wg and blk.first and fuse0:
                dw0 = wg[i][0]
                _hip.check(L.y2_conv0_wgrad_fused(_hip.ptr(ctx.x), _hip.ptr(blk.z), _hip.ptr(blk.scale), _hip.ptr(blk.shift), _hip.ptr(blk.mean), _hip.ptr(blk.invstd),
                                                  _hip.ptr(e.gamma) if blk.has_bn else None, blk.slope, _hip.ptr(sp), cout, _hip.ptr(sums), _hip.ptr(dw0),
                                                  B, h, w, cin, cout, cout, (2 if ctx.frozen else 1) if blk.has_bn else 0, st_w), 'y2_conv0_wgrad_fused')
                return real(dw0)
            if i in wg and blk.first:
                dw0 = wg[i][0]
                _hip.check(L.y2_conv0_wgrad(_hip.ptr(ctx.x), _hip.ptr(dz), _hip.ptr(dw0), B, h, w, cin, cout, cop, st_w), 'y2_conv0_wgrad')
                return real(dw0)
            if blk.first:
                x4 = torch.zeros(B, h, w, 4, dtype=torch.float32, device=dev)
                x4[..., :cin] = ctx.x.permute(0, 2, 3, 1)          # layout conversion only (NCHW plugin input -> NHWC, 4th channel zero)
                dwp = torch.zeros(cop * k * k * 4, dtype=torch.float32, device=dev)
                _hip.check(L.y2_conv_wgrad(_hip.ptr(x4), _hip.ptr(dz), _hip.ptr(dwp), B, h, w, 4, 4, cop, cop, k, st_w), 'y2_conv_wgrad')
                dw4 = _new(dev, cop, 4, k, k)
                _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp), _hip.ptr(dw4), cop, 4, k, st_w), 'y2_unpack_weight_grad')
                return dw4[:e.cout_r, :cin].contiguous()
            tgt, final, zeroed = wg[i][:3]
            dw = None
            if k == 3:
                dw = dest(weight) if (cop == cout and not e.padded) else _new(dev, cop, cin, k, k)
            if fused6:
                got = _hip.conv_wgrad(blk.x, fused6[1], B, h, w, cin, blk.ldx, cop, cop, k, v=blk.wino_v, out=tgt, zeroed=zeroed, native=dw, dz_pre=True)
            else:
                got = _hip.conv_wgrad(blk.x, dz, B, h, w, cin, blk.ldx, cop, cop, k, v=blk.wino_v, out=tgt, zeroed=zeroed, native=dw)     # direct or Winograd, by measurement
            if final:
                return tgt.view(cout, cin, 1, 1)
            if k == 1:
                return real(tgt.view(cop, cin, 1, 1))
            if got is not dw:       # packed [cout][tap][cin] (the direct kernel's GEMM output): one more pass into the gradient's layout
                _hip.check(L.y2_unpack_weight_grad(_hip.ptr(tgt), _hip.ptr(dw), cop, cin, k, st_w), 'y2_unpack_weight_grad')
            return real(dw)

        if side is not None and not e.padded and (not blk.first or i in wg):       # (the rare paths above mix torch-native kernels in: not forked)
            ev = torch.cuda.Event()
            ev.record(main)                           # dz (and everything before it) is complete on the main stream
            with _hip.launch_on(side):
                # Kernels on the side stream, ALLOCATIONS on the main stream (torch's current stream does not change): the step's memory has one
                # allocation stream, so nothing needs record_stream - whose deferred frees kept every activation of a captured step allocated until
                # the capture ended (14 GB instead of ~8 at 416x416) and left the blocks of a shared arena in "pending free" limbo.  What the side
                # stream reads stays referenced from `late` until the main stream has waited for this weight gradient (two layers on).
                side.wait_event(ev)
                gw = weight_grad(_hip.stream())
                done = torch.cuda.Event()
                done.record(side)
            held = (dz, blk.x, blk.wino_v, ctx.x) + ((blk.z, sp) if fuse0 else ()) + ((fused6[1],) if fused6 else ())
            flush_weight_grads(keep=1)
            late.append([weight, gw, done, held])
        else:
            ready(weight, weight_grad(st))
        blk.wino_v = None
        if blk.first and need_dx:
            # gradient with respect to the image: the first layer's data gradient in a 4-channel NHWC space (its 3 -> 4 zero-padded input
            # channels are the OUTPUT channels of this convolution), then the plugin's NCHW layout - a rare path (receptive-field analysis)
            cin4 = (cin + 3) // 4 * 4
            wpad = torch.zeros(cop, cin4, k, k, dtype=torch.float32, device=dev)
            wpad[:cout, :cin] = e.w
            wd = _new(dev, wpad.numel())
            _hip.check(L.y2_pack_weight(_hip.ptr(wpad), _hip.ptr(wd), cop, cin4, k, 1, st), 'y2_pack_weight')
            dx4 = _new(dev, B, h, w, cin4)
            _conv(L, st, dz, wd, dx4, B, h, w, cop, cop, cin4, k, cin4)
            ctx_dx = dx4[..., :cin].permute(0, 3, 1, 2).contiguous()
            grads['__x__'] = ctx_dx
        if not blk.first:
            # data gradient -> the producer's gradient source
            dx = _new(dev, B, h, w, cin)
            ready_ops = ctx.prepared.get(blk.mod)
            if fused6:
                _conv(L, st, None, ready_ops['wd'], dx, B, h, w, cop, cop, cin, k, cin, u=ready_ops['ud'], grad=True,
                      note=lambda kind, m=blk.mod: ctx.used.add((m, 'wd' if kind == 'w' else 'u6d' if kind == '6' else 'ud')), u_eligible=ready_ops['ud_ok'], u6=ready_ops.get('u6d'), pre=fused6[0])
            elif ready_ops is not None:        # rotated / in-out-swapped operands prepared with the forward's (same parameter version)
                # (the fp16 split mode is for activations: its fixed operand scales assume O(1) values, and gradients are 1e-5 and smaller -
                # their fp16 planes would be subnormal; data gradients stay on the fp32 / bf16-split algorithms)
                dg_split = ready_ops['uds'] if _hip.split_mode() == 'bf16' else None
                _conv(L, st, dz, ready_ops['wd'], dx, B, h, w, cop, cop, cin, k, cin, u=ready_ops['ud'], us=dg_split, us_plane=ready_ops['plane'], grad=True,
                      note=lambda kind, m=blk.mod: ctx.used.add((m, 'wd' if kind == 'w' else 'u6d' if kind == '6' else 'ud')), u_eligible=ready_ops['ud_ok'], u6=ready_ops.get('u6d'))
            else:
                wsrc = e.w
                if cop != cout:
                    wpad = bufs[1].get((blk.name, 'wpad'))               # zero rows for the padded output channels: written once, kept
                    if wpad is None or wpad.shape != (cop, cin, k, k):
                        wpad = bufs[1][(blk.name, 'wpad')] = torch.zeros(cop, cin, k, k, dtype=torch.float32, device=dev)
                    _hip.multi([(_hip.MULTI_COPY, wpad[:cout], wsrc)], st)
                    wsrc = wpad
                wd = _new(dev, wsrc.numel())
                _hip.check(L.y2_pack_weight(_hip.ptr(wsrc), _hip.ptr(wd), cop, cin, k, 1, st), 'y2_pack_weight')
                _conv(L, st, dz, wd, dx, B, h, w, cop, cop, cin, k, cin)
            if DEBUG_TAP is not None and dz is not None:
                DEBUG_TAP(blk.name, dz, dx, None, None)
            # route dx
            if blk.name == 'layers3.0':
                dcat = dx                                           # [B,h,w,4*c_pt + c_l2]
                src_full[i_pass] = (dcat, dcat.shape[-1], 0, 1)     # reorg'ed channels first
                src_full[idx[dnn._blocks()[1][-1][0]]] = (dcat, dcat.shape[-1], 4 * c_pt, 0)
            elif blk.name == 'passthrough':
                j = n1 - 1
                src_full[j] = (dx, cin, 0, 0)
            elif i == n1 + 1 and blk.name.startswith('layers2.'):   # first conv of layers2: its input is the pooled layers1[-1]
                src_pool[n1 - 1] = dx
            else:
                prod = i - 1
                if blocks[prod].pool:
                    src_pool[prod] = dx
                else:
                    src_full[prod] = (dx, cin, 0, 0)
        blk.z = None   # free as we go
    # ---- affine-parameter gradients: fp64 sums -> fp32, one launch; straight into the data-parallel bucket slices when there are any
    items, handed = [], []
    gb_all = None
    for prm, off, ln in affine_grads:
        t = buffer_hook(prm) if buffer_hook is not None else None
        if t is None:
            if gb_all is None:
                gb_all = _new(dev, sums_arena.numel())
                items.append((_hip.MULTI_F64_TO_F32, gb_all, sums_arena))
            t = gb_all[off:off + ln]
        else:
            items.append((_hip.MULTI_F64_TO_F32, t, sums_arena[off:off + ln]))
        handed.append((prm, t))
    _hip.multi(items, st)
    for prm, t in handed:
        ready(prm, t)
    flush_weight_grads()
    L.y2_prof_set_tag(0)
    out = [None, grads.get('__x__')]
    for pid in ctx.param_ids:
        out.append(grads.get(pid))
    ctx.blocks = None
    ctx.prepared = None
    ctx.join = None
    return tuple(out)


class DarknetTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dnn, x, *params):
        ctx.need_dx = x.requires_grad          # gradient with respect to the image (receptive_field_analyzer.py:87): one more data-gradient convolution
        return _darknet_fwd(ctx, dnn, x, params, frozen=False)

    @staticmethod
    def backward(ctx, dout):
        return _darknet_bwd(ctx, dout)


class _Tape(object):
    """Stand-in for an autograd ctx when the training graph is run from inside another Function's backward, or without autograd
    at all (StepPlan): carries what forward() leaves for backward()."""
    need_dx = False
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class DarknetEvalGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dnn, x, *params):
        ctx.need_dx = x.requires_grad
        ctx.dnn, ctx.x, ctx.params = dnn, x.detach(), params
        ctx.key = dnn._versions()
        return dnn.forward_nhwc(x.detach())

    @staticmethod
    def backward(ctx, dout):
        if ctx.dnn._versions() != ctx.key:
            raise RuntimeError('model.yolo2: parameters or buffers were modified between an eval-mode forward and its backward')
        tape = _Tape()
        tape.need_dx = ctx.need_dx
        _darknet_fwd(tape, ctx.dnn, ctx.x, ctx.params, frozen=True)      # recompute with frozen BatchNorm statistics, keeping the activations
        return _darknet_bwd(tape, dout)


# ------------------------------------------------------------------------------------------------ head
class DecodeFn(torch.autograd.Function):
    """model.Inference decode (model/__init__.py:122-135) with gradient to the head image."""

    @staticmethod
    def forward(ctx, feature_nhwc, anchors_dev, A):
        L = _hip.lib()
        f = _hip.f32c(feature_nhwc.detach())
        B, rows, cols, ch = f.shape
        E = ch // A
        C = E - 5
        cells = rows * cols
        dev = f.device
        iou = _new(dev, B, cells, A)
        co, sn, mn, mx = (_new(dev, B, cells, A, 2) for _ in range(4))
        _hip.check(L.y2_decode(_hip.ptr(f), _hip.ptr(anchors_dev), B, rows, cols, A, C, _hip.ptr(iou), _hip.ptr(co), _hip.ptr(sn), _hip.ptr(mn), _hip.ptr(mx),
                               None, None, None, _hip.stream()), 'y2_decode')
        logits = f.view(B, cells, A, E)[..., 5:].contiguous() if C > 0 else torch.empty(0, device=dev)
        ctx.save_for_backward(iou, co)
        ctx.shape = (B, rows, cols, A, C)
        ctx.mark_non_differentiable(mn, mx)
        ctx.set_materialize_grads(False)      # (autograd would launch a zero fill per unused / non-differentiable output)
        return iou, co, sn, mn, mx, logits

    @staticmethod
    def backward(ctx, d_iou, d_co, d_sn, d_mn, d_mx, d_logits):
        L = _hip.lib()
        iou, co = ctx.saved_tensors
        B, rows, cols, A, C = ctx.shape
        dev = iou.device
        df = _new(dev, B, rows, cols, A * (5 + C))
        c = lambda t: _hip.f32c(t) if t is not None else None
        if d_iou is None:
            d_iou = torch.zeros_like(iou)
        if d_co is None:
            d_co = torch.zeros_like(co)
        if d_sn is None:
            d_sn = torch.zeros_like(co)
        d_iou, d_co, d_sn = c(d_iou), c(d_co), c(d_sn)
        d_logits = c(d_logits) if (C > 0 and d_logits is not None) else None
        _hip.check(L.y2_decode_bwd(_hip.ptr(iou), _hip.ptr(co), _hip.ptr(d_iou), _hip.ptr(d_co), _hip.ptr(d_sn), _hip.ptr(d_logits),
                                   _hip.ptr(df), B * rows * cols * A, C, _hip.stream()), 'y2_decode_bwd')
        return df, None, None


def inference_forward(inference, feature):
    """Differentiable model.Inference.forward (training): feature is the plugin's NCHW(-view) output with grad."""
    import model
    A = inference.anchors.size(0)
    _feature = feature.permute(0, 2, 3, 1).contiguous()
    anchors = model._device_anchors(inference.anchors, _feature.device)
    iou, co, sn, mn, mx, logits = DecodeFn.apply(_feature, anchors, A)
    return feature, iou, co, sn, mn, mx, (logits if logits.numel() else None)


class RegionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, iou, co, sn, logits, yx_min, yx_max, gt_min, gt_max, gt_cls, anchors_dev, rows, cols, threshold, reducer=None):
        L = _hip.lib()
        dev = iou.device
        B, cells, A = iou.shape
        n = cells * A
        iou, co, sn, yx_min, yx_max = (_hip.f32c(t.detach()) for t in (iou, co, sn, yx_min, yx_max))
        C = 0 if logits is None else logits.shape[-1]
        lg = _hip.f32c(logits.detach()) if C else None
        gt_min, gt_max = _hip.f32c(gt_min), _hip.f32c(gt_max)
        N = gt_min.shape[1]
        cls_i = cls_oh = None
        if C:
            if gt_cls.dim() > 2:
                cls_oh = _hip.f32c(gt_cls)
            else:
                cls_i = gt_cls.to(torch.int64).contiguous()
        best_iou = _new(dev, B, cells, A)
        best_idx = torch.empty(B, cells, A, dtype=torch.int32, device=dev)
        positive = torch.empty(B, cells, A, dtype=torch.uint8, device=dev)
        sums = torch.empty(6, dtype=torch.float64, device=dev)
        out = _new(dev, 5)
        _hip.check(L.y2_region_loss_fwd(_hip.ptr(iou), _hip.ptr(co), _hip.ptr(sn), _hip.ptr(lg), _hip.ptr(yx_min), _hip.ptr(yx_max),
                                        _hip.ptr(gt_min), _hip.ptr(gt_max), _hip.ptr(cls_i), _hip.ptr(cls_oh), _hip.ptr(anchors_dev),
                                        B, rows, cols, A, C, N, float(threshold), _hip.ptr(best_iou), _hip.ptr(best_idx), _hip.ptr(positive),
                                        _hip.ptr(sums), _hip.ptr(out), _hip.stream()), 'y2_region_loss_fwd')
        if SYNC_POSITIVES and cls_i is not None:
            # exact data-parallel parity of the cls term (mean over positives, model/__init__.py:162): global positive count
            npos = sums[5:6]
            if _sum_over_ranks(npos, reducer):
                _hip.check(L.y2_region_loss_finalize(_hip.ptr(sums), float(B * n), 1, _hip.ptr(out), _hip.stream()), 'y2_region_loss_finalize')
        ctx.saved = (iou, co, sn, lg, gt_min, gt_max, cls_i, cls_oh, anchors_dev, best_iou, best_idx, positive, sums)
        ctx.geom = (B, rows, cols, A, C, N, float(threshold))
        ctx.mark_non_differentiable(best_iou, best_idx, positive)
        ctx.set_materialize_grads(False)
        return out, best_iou, best_idx, positive

    @staticmethod
    def backward(ctx, d_out, *_):
        L = _hip.lib()
        iou, co, sn, lg, gt_min, gt_max, cls_i, cls_oh, anchors_dev, best_iou, best_idx, positive, sums = ctx.saved
        B, rows, cols, A, C, N, thr = ctx.geom
        dev = iou.device
        if d_out is None:
            return (None,) * 14
        w = _hip.f32c(d_out)
        d_iou = torch.empty_like(iou)
        d_co, d_sn = torch.empty_like(co), torch.empty_like(sn)
        d_lg = torch.empty_like(lg) if C else None
        _hip.check(L.y2_region_loss_bwd(_hip.ptr(iou), _hip.ptr(co), _hip.ptr(sn), _hip.ptr(lg), _hip.ptr(gt_min), _hip.ptr(gt_max),
                                        _hip.ptr(cls_i), _hip.ptr(cls_oh), _hip.ptr(anchors_dev), B, rows, cols, A, C, N, thr,
                                        _hip.ptr(best_iou), _hip.ptr(best_idx), _hip.ptr(positive), _hip.ptr(sums), _hip.ptr(w),
                                        _hip.ptr(d_iou), _hip.ptr(d_co), _hip.ptr(d_sn), _hip.ptr(d_lg), _hip.stream()), 'y2_region_loss_bwd')
        return (d_iou, d_co, d_sn, d_lg) + (None,) * 10


class LossDict(dict):
    """The five loss terms as the reference returns them (model/__init__.py:165-167: one-element tensors keyed foreground /
    background / center / size [/ cls]); `vector` is the tensor they are slices of - `weighted_total` sums over it with one launch."""
    vector = None


class _LazyDebug(object):
    """The reference's debug dict (model/__init__.py:167: iou, data, positive, negative), consumed by the TensorBoard summaries
    only: every entry is computed at first access instead of costing 7 launches in every training step."""

    def __init__(self, makers):
        self._makers, self._values = makers, {}

    def __getitem__(self, key):
        if key not in self._values:
            self._values[key] = self._makers[key]()
        return self._values[key]

    def __contains__(self, key):
        return key in self._makers

    def __iter__(self):
        return iter(self._makers)

    def __len__(self):
        return len(self._makers)

    def keys(self):
        return self._makers.keys()

    def items(self):
        return [(k, self[k]) for k in self._makers]

    def get(self, key, default=None):
        return self[key] if key in self._makers else default


def loss(anchors, data, pred, threshold):
    """model.loss (model/__init__.py:138-167): returns (dict of 5 one-element tensors with grad, debug dict)."""
    import model
    iou = pred['iou']
    _hip.require_gpu(iou)
    rows, cols = pred['feature'].size()[-2:]
    anchors_dev = model._device_anchors(anchors, iou.device)
    logits = pred.get('logits')
    dev = iou.device
    gt_min, gt_max, gt_cls = (data[k].to(dev) for k in ('yx_min', 'yx_max', 'cls'))
    out, best_iou, best_idx, positive = RegionLossFn.apply(iou, pred['center_offset'], pred['size_norm'], logits, pred['yx_min'], pred['yx_max'],
                                                           gt_min, gt_max, gt_cls, anchors_dev, rows, cols, threshold,
                                                           getattr(pred['feature'], DP_TAG, None))
    result = LossDict(foreground=out[0:1], background=out[1:2], center=out[2:3], size=out[3:4])
    if logits is not None:
        result['cls'] = out[4:5]
    result.vector = out

    def matched():
        # matched ground truth gathered per slot (plumbing for summaries, not on the hot path)
        B = iou.shape[0]
        flat = best_idx.view(B, -1).long()
        _data = {}
        for key, t in (('yx_min', gt_min), ('yx_max', gt_max), ('cls', gt_cls)):
            if t.dim() == 2:
                _data[key] = torch.gather(t, 1, flat).view(*best_idx.shape)
            else:
                _data[key] = torch.gather(t, 1, flat.unsqueeze(-1).expand(-1, -1, t.shape[-1])).view(*best_idx.shape, -1)
        return _data
    debug = _LazyDebug(dict(iou=lambda: best_iou, data=matched, positive=lambda: positive.bool(),
                            negative=lambda: ~positive.bool() & (best_iou < threshold)))
    return result, debug


_HPARAM_DEV = {}


class _WeightedTotalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vector, weights):
        total = torch.empty(1, dtype=torch.float32, device=vector.device)
        _hip.check(_hip.lib().y2_small_dot(_hip.ptr(vector), _hip.ptr(weights), vector.numel(), _hip.ptr(total), _hip.stream()), 'y2_small_dot')
        ctx.weights = weights
        return total

    @staticmethod
    def backward(ctx, g):
        out = torch.empty_like(ctx.weights)
        _hip.check(_hip.lib().y2_small_scale(_hip.ptr(_hip.f32c(g)), _hip.ptr(ctx.weights), out.numel(), _hip.ptr(out), _hip.stream()), 'y2_small_scale')
        return out, None


def weighted_total(loss_, hparam):
    """`sum(loss[key] * hparam[key] for key in loss)` (train.py:348-349) as one launch over the loss vector (a one-element tensor,
    like the reference's sum); any other mapping takes the reference's expression literally."""
    vec = getattr(loss_, 'vector', None)
    keys = list(loss_)
    if vec is None or not vec.is_cuda or keys != ['foreground', 'background', 'center', 'size', 'cls'][:len(keys)]:
        return sum(loss_[key] * hparam[key] for key in loss_)
    w = tuple(float(hparam[k]) for k in keys) + (0.0,) * (vec.numel() - len(keys))
    key = (w, str(vec.device))
    wd = _HPARAM_DEV.get(key)
    if wd is None:
        if len(_HPARAM_DEV) > 64:
            _HPARAM_DEV.clear()
        wd = _HPARAM_DEV[key] = torch.tensor(w, dtype=torch.float32, device=vec.device)
    return _WeightedTotalFn.apply(vec, wd)


# ------------------------------------------------------------------------------------------------ ResNet plugins
class _ROp(object):
    """One recorded operation of the ResNet training forward (conv+BN+ReLU[+residual], or the stem max-pool)."""
    __slots__ = ('kind', 'conv', 'bn', 'x', 'ldx', 'h', 'w', 'ho', 'wo', 'stride', 'pad', 'k', 'cin', 'cout', 'z', 'scale', 'shift', 'mean', 'invstd',
                 'residual', 'y', 'slope', 'first', 'pool')       # pool = (ksize, stride, pad, pad_end) of a 'pool' op


def resnet_forward(net, x, frozen=False):
    """frozen: eval()-mode BatchNorm (running statistics, nothing updated) with autograd recording."""
    params = [p for p in net.parameters()]
    out = ResNetTrainFn.apply(net, x, frozen, *params)
    return out.permute(0, 3, 1, 2)


def tiny_forward(net, x, frozen=False):
    """Training-mode forward of model.yolo2.Tiny (model/yolo2.py:140-173) through the same op-list graph as the ResNets."""
    params = [p for p in net.parameters()]
    out = ResNetTrainFn.apply(net, x, frozen, *params)
    return out.permute(0, 3, 1, 2)


def _gen_conv(L, st, x, wp, y, B, H, W, cin, ldx, cout, k, stride, pad, stats=None, transposed=False, out_hw=None):
    p = _hip.ConvParams()
    p.x, p.w, p.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    p.stats = stats.data_ptr() if stats is not None else None
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, H, W, cin, ldx, cout, k
    p.ldy, p.slope, p.tile = cout, 1.0, 0
    p.stride, p.pad_plus1 = stride, pad + 1
    if transposed:
        p.transposed, p.out_h, p.out_w = 1, out_hw[0], out_hw[1]
    u = _hip.wino_weight(wp, cout, cin) if (not transposed and stride == 1 and pad == 1 and _hip.wino_eligible(cout, cin, k)) else None
    _hip.autotune_conv(p, x.device, wino_w=u)
    _hip.conv_workspace(p, x.device)
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), st), 'y2_conv_fwd')


def _resnet_operands(net, dev, scope=None):
    """{nn.Conv2d: dict(wp, wd)}: the forward / data-gradient GEMM operands of every convolution of a ResNet / Tiny plugin whose channel
    counts need no padding, derived by ONE y2_prep_weights launch per parameter version (the per-layer path costs two y2_pack_weight
    launches per convolution and step: 107 for ResNet-50).  scope: see _train_operands."""
    import torch.nn as nn
    convs = [m for m in net.modules() if isinstance(m, nn.Conv2d)]
    key = (dev, tuple((c.weight.data_ptr(), c.weight._version) for c in convs))
    if scope is not None:
        bufs = scope
    else:
        cache = net.__dict__.get('_train_cache')
        if cache is not None and cache[0] == key:
            return cache[1]
        held = net.__dict__.get('_train_bufs')
        if held is None or held[0] != dev:
            held = net.__dict__['_train_bufs'] = (dev, {})
        bufs = held[1]
    items, ops = [], {}
    for i, c in enumerate(convs):
        w = c.weight.detach()
        cout, cin, k, _ = w.shape
        if cout % 4 or cin % 4 or not w.is_contiguous() or w.dtype != torch.float32 or not w.is_cuda:
            continue          # the 3-channel stem and the 425-wide head run zero-padded: per-layer path
        d = {}
        for tag, mode in (('wp', _hip.PREP_FPROP), ('wd', _hip.PREP_DGRAD)):
            t = bufs.get(('rn', i, tag))
            if t is None or t.numel() != w.numel():
                t = bufs[('rn', i, tag)] = torch.empty(w.numel(), dtype=torch.float32, device=dev)
            d[tag] = t
            items.append((w, t, cout, cin, k, mode))
        ops[c] = d
    if items:
        table = (_hip.PrepItem * len(items))()
        for e, (src, dst, cout, cin, k, mode) in zip(table, items):
            e.src, e.dst, e.Cout, e.Cin, e.ksize, e.mode = src.data_ptr(), dst.data_ptr(), cout, cin, k, mode
        _hip.check(_hip.lib().y2_prep_weights(table, len(items), _hip.stream()), 'y2_prep_weights')
    if scope is None:
        net.__dict__['_train_cache'] = (key, ops)
    return ops


class ResNetTrainFn(torch.autograd.Function):
    """Training graph of model.resnet.ResNet (model/resnet.py:29-158): per convolution {raw general conv with BN statistics in the
    epilogue -> y2_bn_finalize (momentum 0.1) -> y2_bn_act_fwd_ex (affine [+ residual] + ReLU)}; backward in reverse with
    gradient fan-in per tensor: y2_bn_act_bwd_ex (ReLU mask from the recomputed pre-activation, BN backward, gradient of
    the residual input) -> y2_conv_wgrad_ex -> data gradient (stride 1: forward kernel on rotated weights; stride 2:
    transposed mode of the general kernel); the stem max-pool goes through y2_maxpool_fwd / y2_maxpool_bwd."""
    MOMENTUM = 0.1

    @staticmethod
    def forward(ctx, net, x, frozen, *params):
        _hip.require_gpu(x)
        ctx.need_dx = x.requires_grad
        L = _hip.lib()
        st = _hip.stream()
        x = _hip.f32c(x.detach())
        B, cin0, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError('input size must be a multiple of 32 (got %dx%d)' % (H, W))
        dev = x.device
        ops = []
        ctx.frozen = frozen
        scope = getattr(ctx, 'scope', None)
        prepared = ctx.prepared = _resnet_operands(net, dev, scope)
        ctx.prepared_key = net.__dict__['_train_cache'][0] if scope is None else None
        import torch.nn as nn
        # one zero-filled arena for the replicated BatchNorm-statistics accumulators of every convolution (one launch instead of 53 fills)
        arena = None
        if not frozen:
            arena = torch.empty(_hip.STATS_REPL * 2 * sum(m.num_features for m in net.modules() if isinstance(m, nn.BatchNorm2d)), dtype=torch.float64, device=dev)
            if arena.numel():
                _hip.multi([(_hip.MULTI_ZERO, arena, None)])
        used = [0]
        cpad = (cin0 + 3) // 4 * 4
        x4 = _new(dev, B, H, W, cpad)
        _hip.check(L.y2_nchw_to_nhwc(_hip.ptr(x), _hip.ptr(x4), B, cin0, H, W, cpad, st), 'y2_nchw_to_nhwc')
        ctx.x4, ctx.cin0 = x4, cin0

        def conv_bn(conv, bn, xin, ldx, h, w, stride, pad, slope, residual=None, first=False, momentum=None):
            op = _ROp()
            weight = _hip.f32c(conv.weight.detach())
            cout, cin_true, k, _ = weight.shape
            if ldx % 4 or (cin_true % 4 and not first):
                raise RuntimeError('training needs conv input channel counts that are multiples of 4 (got %d)' % cin_true)
            if cin_true != ldx:           # stem: zero-padded input channels
                wpad = torch.zeros(cout, ldx, k, k, dtype=torch.float32, device=dev)
                wpad[:, :cin_true] = weight
                weight = wpad
            if conv in prepared and cin_true == ldx:
                wp = prepared[conv]['wp']
            else:
                wp = _new(dev, weight.numel())
                _hip.check(L.y2_pack_weight(_hip.ptr(weight), _hip.ptr(wp), cout, ldx, k, 0, st), 'y2_pack_weight')
            ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
            z = _new(dev, B, ho, wo, cout)
            stats = None
            if bn is not None and not frozen:
                stats = arena[used[0]:used[0] + _hip.STATS_REPL * 2 * cout]
                used[0] += _hip.STATS_REPL * 2 * cout
            det = _hip.ensure_deterministic(dev)
            _gen_conv(L, st, xin, wp, z, B, h, w, ldx, ldx, cout, k, stride, pad, stats=None if det else stats)
            if det and stats is not None:
                _hip.colstats_det(z, B * ho * wo, cout, cout, stats)
            op.kind, op.conv, op.bn, op.x, op.ldx, op.h, op.w, op.ho, op.wo = 'conv', conv, bn, xin, ldx, h, w, ho, wo
            op.stride, op.pad, op.k, op.cin, op.cout, op.z, op.residual, op.slope, op.first = stride, pad, k, cin_true, cout, z, residual, slope, first
            if bn is not None and frozen:
                op.scale, op.shift = _new(dev, cout), _new(dev, cout)
                _hip.check(L.y2_bn_fold(_hip.ptr(_hip.f32c(bn.weight.detach())), _hip.ptr(_hip.f32c(bn.bias.detach())), _hip.ptr(_hip.f32c(bn.running_mean)),
                                        _hip.ptr(_hip.f32c(bn.running_var)), BN_EPS, _hip.ptr(op.scale), _hip.ptr(op.shift), cout, st), 'y2_bn_fold')
                op.mean, op.invstd = _hip.f32c(bn.running_mean), torch.rsqrt(_hip.f32c(bn.running_var) + BN_EPS)
            elif bn is not None:
                op.scale, op.shift, op.mean, op.invstd = (_new(dev, cout) for _ in range(4))
                _hip.check(L.y2_bn_finalize(_hip.ptr(stats), float(B * ho * wo), _hip.ptr(bn.weight.detach()), _hip.ptr(bn.bias.detach()),
                                            _hip.ptr(bn.running_mean), _hip.ptr(bn.running_var), ResNetTrainFn.MOMENTUM if momentum is None else momentum, BN_EPS,
                                            _hip.ptr(op.scale), _hip.ptr(op.shift), _hip.ptr(op.mean), _hip.ptr(op.invstd), cout,
                                            _hip.ptr(_counter(bn)), st), 'y2_bn_finalize')
                _hip.wrote([t for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked) if t is not None])      # written through raw pointers
            else:
                op.scale = op.mean = op.invstd = None
                op.shift = _hip.f32c(conv.bias.detach()) if conv.bias is not None else None
            y = _new(dev, B, ho, wo, cout)
            _hip.check(L.y2_bn_act_fwd_ex(_hip.ptr(z), _hip.ptr(op.scale), _hip.ptr(op.shift), slope, _hip.ptr(residual), cout if residual is not None else 0,
                                          _hip.ptr(y), None, B, ho, wo, cout, cout, cout, 0, 0, 0, 0, st), 'y2_bn_act_fwd_ex')
            op.y = y
            ops.append(op)
            return y, ho, wo, cout

        def maxpool(cur, h, w, ld, ksize, stride, pad, pad_end):
            pool = _ROp()
            pool.kind, pool.x, pool.h, pool.w, pool.cout, pool.pool = 'pool', cur, h, w, ld, (ksize, stride, pad, pad_end)
            ph, pw = (h + pad + pad_end - ksize) // stride + 1, (w + pad + pad_end - ksize) // stride + 1
            pooled = _new(dev, B, ph, pw, ld)
            _hip.check(L.y2_maxpool_fwd(_hip.ptr(cur), _hip.ptr(pooled), B, h, w, ld, ld, ld, ksize, stride, pad, pad_end, st), 'y2_maxpool_fwd')
            pool.y, pool.ho, pool.wo = pooled, ph, pw
            ops.append(pool)
            return pooled, ph, pw

        from model import yolo2 as _yolo2
        if isinstance(net, _yolo2.Tiny):
            # model/yolo2.py:140-173: nn.Sequential of Conv2d blocks (BN momentum 0.01, LeakyReLU 0.1), MaxPool2d(2) and the
            # ConstantPad2d((0,1,0,1)) + MaxPool2d(2, stride=1) pair; the 3-channel input runs zero-padded to 4 NHWC channels
            cur, h, w, ld = x4, H, W, cpad
            mods = list(net.layers)
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, _yolo2.Conv2d):
                    cur, h, w, ld = conv_bn(m.conv, m.bn, cur, ld, h, w, 1, (m.kernel_size - 1) // 2, LEAKY if m.has_act else 1.0,
                                            first=(i == 0), momentum=BN_MOMENTUM)
                elif isinstance(m, _yolo2._PadPool):
                    cur, h, w = maxpool(cur, h, w, ld, 2, 1, 0, 1)
                    i += 1          # the pad + pool pair
                else:
                    cur, h, w = maxpool(cur, h, w, ld, 2, 2, 0, 0)
                i += 1
            ctx.net, ctx.ops, ctx.B = net, ops, B
            ctx.param_ids = [id(p) for p in params]
            return cur
        cur, h, w, ld = conv_bn(net.conv1, net.bn1, x4, cpad, H, W, 2, 3, 0.0, first=True)
        cur, h, w = maxpool(cur, h, w, ld, 3, 2, 1, 1)
        for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
            for blk in layer:
                residual = cur
                if blk.downsample is not None:
                    residual, _, _, _ = conv_bn(blk.downsample[0], blk.downsample[1], cur, ld, h, w, blk.stride, 0, 1.0)
                t, th, tw, tld = cur, h, w, ld
                convs = blk.convs()
                for i, (conv, bn, cs, cp) in enumerate(convs):
                    last = i == len(convs) - 1
                    t, th, tw, tld = conv_bn(conv, bn, t, tld, th, tw, cs, cp, 0.0, residual=residual if last else None)
                cur, h, w, ld = t, th, tw, tld
        out, _, _, _ = conv_bn(net.conv, None, cur, ld, h, w, 1, 0, 1.0)
        ctx.net, ctx.ops, ctx.B = net, ops, B
        ctx.param_ids = [id(p) for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _hip.lib()
        st = _hip.stream()
        net, ops, B = ctx.net, ctx.ops, ctx.B
        dev = dout.device
        prepared = getattr(ctx, 'prepared', None) or {}
        if prepared and getattr(ctx, 'prepared_key', None) is not None and net.__dict__.get('_train_cache', (None,))[0] != ctx.prepared_key:
            raise RuntimeError('model.resnet: a convolution weight was modified between this forward and its backward; the per-model GEMM operand '
                               'buffers this graph was recorded against hold other weights now')
        grads = {}
        hook = getattr(net, 'grad_ready_hook', None)
        buffer_hook = getattr(net, 'grad_buffer_hook', None)

        def ready(param, g):
            grads[id(param)] = g
            if hook is not None:
                hook(param, g)

        def dest(param):
            t = buffer_hook(param) if buffer_hook is not None else None
            return t if t is not None else _new(dev, *param.shape)
        convs = [op for op in ops if op.kind == 'conv']
        # everything that must start from zero, filled by ONE launch: the fp64 sums of every BatchNorm backward, the accumulation targets of the
        # direct (split, atomically added) weight gradients, the zero-padded gradient of the 425-wide head
        sums_arena = torch.empty(2 * sum(op.cout for op in convs), dtype=torch.float64, device=dev)
        zero = [sums_arena]
        plan = {}
        off = 0
        for op in convs:
            cout, cin, k = op.cout, op.ldx, op.k
            cop = (cout + 3) // 4 * 4
            e = plan[id(op)] = dict(sums=sums_arena[off:off + 2 * cout], off=off)
            off += 2 * cout
            e['dz'] = None
            if cop != cout:
                e['dz'] = _new(dev, B, op.ho, op.wo, cop)
                zero.append(e['dz'])
            wino = k == 3 and op.stride == 1 and op.pad == 1
            e['wino'] = wino
            if not wino:
                # [cop][k*k][cin]: for a 1x1 convolution that IS the state_dict layout - the kernel writes the gradient tensor itself
                direct_out = k == 1 and cop == cout and cin == op.cin
                e['dwp'] = dest(op.conv.weight).view(-1) if direct_out else _new(dev, cop * k * k * cin)
                e['final'] = direct_out
                zero.append(e['dwp'])
        _hip.multi([(_hip.MULTI_ZERO, t, None) for t in zero], st)
        affine = []
        G = {id(ops[-1].y): [_hip.f32c(dout)]}      # gradient sources per activation tensor
        for op in reversed(ops):
            srcs = G.pop(id(op.y), [])
            assert 1 <= len(srcs) <= 2, len(srcs)
            if op.kind == 'pool':
                dx = _new(dev, B, op.h, op.w, op.cout)
                pk, ps, pp, pe = op.pool
                _hip.check(L.y2_maxpool_bwd(_hip.ptr(op.x), _hip.ptr(srcs[0]), _hip.ptr(srcs[1]) if len(srcs) > 1 else None, _hip.ptr(dx),
                                            B, op.h, op.w, op.cout, op.cout, op.cout, op.cout, pk, ps, pp, pe, st), 'y2_maxpool_bwd')
                G.setdefault(id(op.x), []).append(dx)
                continue
            cout, cin, k, ho, wo = op.cout, op.ldx, op.k, op.ho, op.wo
            cop = (cout + 3) // 4 * 4
            e = plan[id(op)]
            sums, dz = e['sums'], (e['dz'] if e['dz'] is not None else _new(dev, B, ho, wo, cop))
            dres = _new(dev, B, ho, wo, cout) if op.residual is not None else None
            has_bn = op.bn is not None
            _hip.check(L.y2_bn_act_bwd_ex(_hip.ptr(op.z), _hip.ptr(op.scale), _hip.ptr(op.shift), _hip.ptr(op.mean), _hip.ptr(op.invstd),
                                          _hip.ptr(op.bn.weight.detach()) if has_bn else None, op.slope,
                                          _hip.ptr(srcs[0]), cout, 0, 0, None, 0, 0,
                                          _hip.ptr(srcs[1]) if len(srcs) > 1 else None, cout,
                                          _hip.ptr(op.residual), cout if op.residual is not None else 0, _hip.ptr(dres), cout,
                                          _hip.ptr(sums), _hip.ptr(dz), cop, B, ho, wo, cout, cout, (2 if ctx.frozen else 1) if has_bn else 0, st), 'y2_bn_act_bwd_ex')
            if op.residual is not None:
                G.setdefault(id(op.residual), []).append(dres)
            # parameter gradients of the affine part = the fp64 sums of pass 1: converted for ALL layers by one launch after the loop
            if has_bn:
                affine.append((op.bn.bias, e['off'], cout))
                affine.append((op.bn.weight, e['off'] + cout, cout))
            elif op.conv.bias is not None:
                affine.append((op.conv.bias, e['off'], cout))
            # ---- weight gradient
            if e['wino']:
                dwp = _hip.conv_wgrad(op.x, dz, B, op.h, op.w, cin, cin, cop, cop, k)     # direct or Winograd, by measurement
            else:
                dwp = e['dwp']
                _hip.check(L.y2_conv_wgrad_ex(_hip.ptr(op.x), _hip.ptr(dz), _hip.ptr(dwp), B, op.h, op.w, cin, cin, cop, cop, k, op.stride, op.pad, st), 'y2_conv_wgrad_ex')
            if not e['wino'] and e['final']:
                ready(op.conv.weight, dwp.view(cout, cin, 1, 1))
            else:
                dw = dest(op.conv.weight) if (cop == cout and cin == op.cin) else _new(dev, cop, cin, k, k)
                _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp), _hip.ptr(dw), cop, cin, k, st), 'y2_unpack_weight_grad')
                ready(op.conv.weight, dw if (cop == cout and cin == op.cin) else dw[:cout, :op.cin].contiguous())
            op.z = None
            if op.first and not ctx.need_dx:
                continue
            # ---- data gradient (of the first layer only when the image's gradient is wanted: its result is the 4-channel NHWC image gradient)
            ready_ops = prepared.get(op.conv)
            if ready_ops is not None and cop == cout and op.cin == cin:
                wd = ready_ops['wd']          # rotated / in-out-swapped operand prepared with the forward's (same parameter version)
            else:
                wsrc = _hip.f32c(op.conv.weight.detach())
                if cop != cout or wsrc.shape[1] != cin:          # zero rows for padded output channels, zero columns for the stem's padded input channels
                    wpad = torch.zeros(cop, cin, k, k, dtype=torch.float32, device=dev)
                    wpad[:cout, :wsrc.shape[1]] = wsrc
                    wsrc = wpad
                wd = _new(dev, wsrc.numel())
                _hip.check(L.y2_pack_weight(_hip.ptr(wsrc), _hip.ptr(wd), cop, cin, k, 1, st), 'y2_pack_weight')
            dx = _new(dev, B, op.h, op.w, cin)
            if op.stride == 1:
                _gen_conv(L, st, dz, wd, dx, B, ho, wo, cop, cop, cin, k, 1, k - 1 - op.pad)
            else:
                _gen_conv(L, st, dz, wd, dx, B, ho, wo, cop, cop, cin, k, op.stride, op.pad, transposed=True, out_hw=(op.h, op.w))
            G.setdefault(id(op.x), []).append(dx)
        # ---- affine-parameter gradients: fp64 sums -> fp32, one launch; straight into the data-parallel bucket slices when there are any
        items, handed, gb_all = [], [], None
        for prm, o, ln in affine:
            t = buffer_hook(prm) if buffer_hook is not None else None
            if t is None:
                if gb_all is None:
                    gb_all = _new(dev, sums_arena.numel())
                    items.append((_hip.MULTI_F64_TO_F32, gb_all, sums_arena))
                t = gb_all[o:o + ln]
            else:
                items.append((_hip.MULTI_F64_TO_F32, t, sums_arena[o:o + ln]))
            handed.append((prm, t))
        _hip.multi(items, st)
        for prm, t in handed:
            ready(prm, t)
        dx_img = None
        if ctx.need_dx:
            gx = G.pop(id(ctx.x4), None)
            if gx:
                dx_img = gx[0][..., :ctx.cin0].permute(0, 3, 1, 2).contiguous()
        out = [None, dx_img, None]
        for pid in ctx.param_ids:
            out.append(grads.get(pid))
        ctx.ops = None
        ctx.prepared = None
        return tuple(out)


# ------------------------------------------------------------------------------------------------ a training step as a plan
# The reference's step is three Python calls - `_inference`, `loss`, `loss_total.backward()` (train.py:344-351) - and on the autograd
# path above every one of the ~270 kernels behind them is a ctypes call issued from Python, one by one, every step: 6-18 ms of host
# time per step, more than half of the GPU time of a 320x320 step, and with eight ranks sharing one host the part that scales worst.
# A StepPlan runs the SAME launch sequence (the forward / backward bodies of the Functions above, called directly with _Tape objects:
# no autograd bookkeeping) on static input buffers, and captures it once per problem shape into hipGraph segments; a later step is a
# copy into the input buffers plus one graph launch per segment.  Segments end where the data-parallel wrapper has a collective to
# issue (the positive-count sum inside the loss, every gradient bucket that completes during backward): RCCL calls stay ordinary
# eager calls between two graph launches, in the same order and with the same payloads as on the autograd path, so ranks in graph
# mode and ranks still warming up on another input size interoperate.
class _Segments(object):
    """Capture state of one StepPlan pass: a list of ops - ('graph', CUDAGraph) | ('npos', tensor) | ('buckets', lo, hi) - in issue order."""

    def __init__(self, pool, mode):
        self.pool, self.mode = pool, mode
        self.ops, self.graph, self.mark = [], None, 0
        self.held = []

    def begin(self):
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: calls other threads make meanwhile (a pinned-memory loader thread, RCCL's watchdog) must not abort the capture
        self.graph.capture_begin(pool=self.pool, capture_error_mode='thread_local')
        self.mark = _hip.LAUNCHES[0]

    def cut(self, op, maybe_foreign=False):
        """A collective belongs here.  A segment without any launch is not closed (hipGraphInstantiate of nothing): the op then simply follows
        its predecessor.  _hip.LAUNCHES counts the LIBRARY's launches only; a caller that may have enqueued torch-native kernels since the last
        cut says so (maybe_foreign): the segment is then closed behind one trivial library launch, so that nothing it holds can be reordered
        behind the collective."""
        if _hip.LAUNCHES[0] == self.mark and maybe_foreign:
            t = torch.empty(4, dtype=torch.float32, device=torch.device('cuda', torch.cuda.current_device()))
            self.held.append(t)            # lives as long as the graphs that write it (the plan keeps `held`)
            _hip.multi([(_hip.MULTI_ZERO, t, None)])
        if _hip.LAUNCHES[0] != self.mark:
            self.graph.capture_end()
            self.ops.append(('graph', self.graph))
            self.ops.append(op)
            self.begin()
        else:
            self.ops.append(op)

    def end(self):
        self.graph.capture_end()
        if _hip.LAUNCHES[0] != self.mark:
            self.ops.append(('graph', self.graph))
        self.graph = None

    def abort(self, fork=None):
        """End a capture that failed half-way.  fork: the stream the backward forks weight gradients onto - when the failure came between a
        fork and its join that stream is part of this capture, and a capture cannot end with an unjoined branch (it would stay in capture
        mode and poison the next attempt): join it first."""
        if self.graph is not None:
            try:
                if fork is not None:
                    with torch.cuda.stream(fork):
                        forked = torch.cuda.is_current_stream_capturing()
                    if forked:
                        torch.cuda.current_stream().wait_stream(fork)
                self.graph.capture_end()
            except Exception:
                pass
            self.graph = None


class StepPlan(object):
    """fwd + region loss + bwd of one training step for ONE problem shape (per-GPU batch, input size, padded box count, label form).

    run(data)  -> dict(pred, loss, loss_total, debug) like train.iterate's; afterwards every parameter's .grad holds this step's
                  gradient (averaged over the ranks under a DataParallelRCCL wrapper), written in place of - never added to - what was there.
    The first `WARM` calls run the launch sequence eagerly (per-layer algorithm measurements happen there), the next one captures it,
    every later one replays.  Results are views of static buffers: valid until the next step."""
    WARM = 3

    def __init__(self, inference, anchors, hparam, threshold, dp=None, pool=None, shared=None, arena=None, scope=None):
        """arena: a torch.cuda.MemPool owned by the caller (train.StepRunner) - the ONE activation arena of all its plans: the eager warm-up passes allocate from it
        too (on the capture stream: the allocator reuses a freed block only on the stream that allocated it), so a plan's intermediates live in the memory the
        previous plan's did (plans run strictly one after the other) instead of in a pool of their own next to the default allocator's cached copy of the same.
        shared: a dict owned by the caller (train.StepRunner) for the prepared GEMM-operand buffers: plans replay strictly one after the
        other and every replay rewrites the operands it reads at its head, so the plans of all input sizes can use ONE set of buffers
        (ten multi-scale sizes would otherwise hold ten copies: ~1 GB each)."""
        import model
        from model import yolo2 as _yolo2
        self.inference, self.dnn, self.dp = inference, inference.dnn, dp
        self.anchors, self.hparam, self.threshold = anchors, dict(hparam), float(threshold)
        self.darknet = isinstance(self.dnn, _yolo2.Darknet) and not isinstance(self.dnn, _yolo2.Tiny)
        self.params = [p for p in self.dnn.parameters()]
        self.buffers = [b for b in self.dnn.buffers()]
        self.arena = arena
        self.pool = arena.id if arena is not None else (pool if pool is not None else torch.cuda.graph_pool_handle())       # all segments (and, shared by the runner, all shapes) allocate from one pool
        # scope: what a captured step holds by address besides its intermediates - kernel scratch (_hip.SCOPE), the packed staging of the 3x3 weight gradients,
        # cached constants.  None: private to this plan.  The runner of an arena hands ONE dict to all its plans: none of it depends on the input size except
        # the scratch, which only ever grows (a replaced buffer stays alive in the dict: _hip._retire) - ten sizes hold one set, not ten
        self.scope = scope if scope is not None else {}
        self.shared = shared if shared is not None else self.scope
        self.used_last = None       # (block, operand form) pairs the last eager pass read: what the captured step prepares
        self.only = None
        self.ops = None             # captured op list
        self.last_grads = {}
        self.capture_error = None   # what a failed capture raised (the plan then stays on eager launches)
        self.calls = 0
        self.static = None
        self.result = None
        self.grads = {}

    # ---- static inputs
    def _alloc(self, data, npad):
        x = data['tensor']
        dev = x.device
        B, _, H, W = x.shape
        cls = data['cls']
        st = dict(x=torch.empty(B, x.shape[1], H, W, dtype=torch.float32, device=dev),
                  gt_min=torch.zeros(B, npad, 2, dtype=torch.float32, device=dev), gt_max=torch.zeros(B, npad, 2, dtype=torch.float32, device=dev),
                  cls=torch.zeros((B, npad) + tuple(cls.shape[2:]), dtype=torch.int64 if cls.dim() == 2 else torch.float32, device=dev),
                  npad=npad)
        self.static = st

    def _load(self, data):
        """Copy a batch into the static inputs: image as it is, boxes in grid-cell units (train.norm_data: pixels x rows/height,
        cols/width), rows past this batch's box count zero = the invalid boxes the collate function pads with (utils/data.py:114-133)."""
        st = self.static
        st['x'].copy_(data['tensor'], non_blocking=True)
        H, W = st['x'].shape[-2:]
        rows, cols = H // 32, W // 32
        n = data['yx_min'].shape[1]
        scale = _scale_tensor(rows / H, cols / W, st['x'].device)
        for key, buf in (('yx_min', st['gt_min']), ('yx_max', st['gt_max'])):
            v = buf[:, :n]
            v.copy_(data[key], non_blocking=True)
            v.mul_(scale)
            if n < st['npad']:
                buf[:, n:].zero_()
        st['cls'][:, :n].copy_(data['cls'], non_blocking=True)
        if n < st['npad']:
            st['cls'][:, n:].zero_()

    # ---- the launch sequence
    def _chain(self, seg, grads):
        """Issue every launch of the step on the current stream.  seg: None (eager pass: collectives run where they belong) or a
        _Segments being captured.  grads: dict filled with {id(param): gradient tensor}."""
        import model
        dnn, dp, st = self.dnn, self.dp, self.static
        dev = st['x'].device
        anchors_dev = model._device_anchors(self.anchors, dev)
        A = self.anchors.size(0)
        nb = len(dp._buckets) if dp is not None else 0
        ready_n = [0] * nb
        state = dict(next=0)

        def dest(p):
            if dp is not None:
                return dp.graph_slot(p)
            t = grads.get(('dest', id(p)))
            if t is None:
                # with an arena the gradient tensors are the runner's, one set for the plans of all sizes (p.grad is the same tensor whatever size ran last)
                home = self.scope if self.arena is not None else grads
                t = home.get(('dest', id(p)))
                if t is None:
                    t = home[('dest', id(p))] = torch.empty_like(p, memory_format=torch.contiguous_format)
                grads[('dest', id(p))] = t
            return t

        def ready(p, g):
            if dp is None:
                grads[id(p)] = g
                return
            slot = dp.graph_slot(p)
            if slot is None:
                raise RuntimeError('StepPlan: a parameter of the plugin is not in the data-parallel wrapper')
            if g.data_ptr() != slot.data_ptr() or not g.is_contiguous():
                # (a copy KERNEL: torch's contiguous device-to-device copy_ is a memcpy node under capture, and memset nodes already proved to run at a
                # graph's first launch only on this runtime - nothing of that kind goes into a captured step)
                _hip.multi([(_hip.MULTI_COPY, slot.view(-1), _hip.f32c(g).view(-1))])
            if id(p) in grads:
                return
            grads[id(p)] = slot
            bi = dp._where[id(p)][0]
            ready_n[bi] += 1
            lo = state['next']
            hi = lo
            while hi < nb and ready_n[hi] == len(dp._buckets[hi]):
                hi += 1
            if hi > lo and hi < nb:          # (the last bucket goes out after the last segment: nothing is left to overlap it with)
                state['next'] = hi
                if seg is None:
                    dp.graph_launch(lo, hi)
                else:
                    join = getattr(state['tape'], 'join', None)
                    if join is not None:
                        join()               # weight gradients still running on the backward's side stream: a graph segment ends with every forked stream joined
                    seg.cut(('buckets', lo, hi), maybe_foreign=True)

        def npos(t):
            if seg is None:
                return dp._sum_small(t)
            seg.cut(('npos', t), maybe_foreign=True)
            return True

        hooks = (getattr(dnn, 'grad_ready_hook', None), getattr(dnn, 'grad_buffer_hook', None))
        dnn.grad_ready_hook, dnn.grad_buffer_hook = ready, dest
        try:
            tape = state['tape'] = _Tape()
            tape.fork_ok = (dp is None) if GRAPH_FORK == 'auto' else bool(GRAPH_FORK)
            if self.darknet:
                tape.ops_scope, tape.only = (self.shared, self.only) if seg is not None else (None, None)
                head = _darknet_fwd(tape, dnn, st['x'], self.params, False, scope=self.scope if seg is not None else None)
            else:
                tape.scope = self.scope if seg is not None else None
                head = ResNetTrainFn.forward(tape, dnn, st['x'], False, *self.params)
            B, rows, cols, _ = head.shape
            dt, lt = _Tape(), _Tape()
            iou, co, sn, mn, mx, logits = DecodeFn.forward(dt, head, anchors_dev, A)
            logits = logits if logits.numel() else None
            reducer = npos if (dp is not None and dp.world > 1) else None
            out, best_iou, best_idx, positive = RegionLossFn.forward(lt, iou, co, sn, logits, mn, mx, st['gt_min'], st['gt_max'], st['cls'], anchors_dev,
                                                                     rows, cols, self.threshold, reducer)
            keys = ['foreground', 'background', 'center', 'size'] + (['cls'] if logits is not None else [])
            w = _hparam_tensor(tuple(float(self.hparam[k]) for k in keys) + (0.0,) * (out.numel() - len(keys)), dev)
            self.scope['held'] = (w, anchors_dev)          # (cache entries other code may drop: a captured graph reads them at every replay)
            total = torch.empty(1, dtype=torch.float32, device=dev)
            _hip.check(_hip.lib().y2_small_dot(_hip.ptr(out), _hip.ptr(w), out.numel(), _hip.ptr(total), _hip.stream()), 'y2_small_dot')
            # ---- backward: d total / d total = 1, so the loss vector's gradient IS the weight vector (what _WeightedTotalFn.backward computes)
            d_iou, d_co, d_sn, d_lg = RegionLossFn.backward(lt, w)[:4]
            df = DecodeFn.backward(dt, d_iou, d_co, d_sn, None, None, d_lg)[0]
            if self.darknet:
                _darknet_bwd(tape, df)
            else:
                ResNetTrainFn.backward(tape, df)
        finally:
            dnn.grad_ready_hook, dnn.grad_buffer_hook = hooks
        missing = [p for p in self.params if p.requires_grad and id(p) not in grads]
        if missing:
            raise RuntimeError('StepPlan: %d parameters received no gradient' % len(missing))
        if self.darknet and seg is None:
            self.used_last = set(getattr(tape, 'used', ()))
        pred = dict(feature=head.permute(0, 3, 1, 2), iou=iou, center_offset=co, size_norm=sn, yx_min=mn, yx_max=mx)
        if logits is not None:
            pred['logits'] = logits
        result = LossDict(foreground=out[0:1], background=out[1:2], center=out[2:3], size=out[3:4])
        if logits is not None:
            result['cls'] = out[4:5]
        result.vector = out
        gt_min, gt_max, gt_cls, thr = st['gt_min'], st['gt_max'], st['cls'], self.threshold

        def matched():
            flat = best_idx.view(B, -1).long()
            _data = {}
            for key, t in (('yx_min', gt_min), ('yx_max', gt_max), ('cls', gt_cls)):
                if t.dim() == 2:
                    _data[key] = torch.gather(t, 1, flat).view(*best_idx.shape)
                else:
                    _data[key] = torch.gather(t, 1, flat.unsqueeze(-1).expand(-1, -1, t.shape[-1])).view(*best_idx.shape, -1)
            return _data
        debug = _LazyDebug(dict(iou=lambda: best_iou, data=matched, positive=lambda: positive.bool(),
                                negative=lambda: ~positive.bool() & (best_iou < thr)))
        return dict(pred=pred, loss=result, loss_total=total, debug=debug), state['next']

    # ---- one step
    def _finish(self, grads, launched):
        """Outstanding collectives, averaging, and the parameters' .grad."""
        dp = self.dp
        if dp is not None:
            dp.graph_launch(launched, len(dp._buckets))
            dp.graph_finish()
        for p in self.params:
            g = grads.get(id(p))
            if g is not None:
                p.grad = g
        self.last_grads = grads
        if self.buffers:
            _hip.wrote(self.buffers)          # BatchNorm running statistics / step counters were updated by y2_bn_finalize (raw pointers)

    def run(self, data, capture=True):
        self.calls += 1
        if self.dp is not None:
            self.dp.graph_begin()
        if self.static is None:
            raise RuntimeError('StepPlan.run before StepPlan._alloc')
        self._load(data)
        if self.ops is None and capture and self.calls > self.WARM and self.capture_error is None:
            try:
                self._capture()
            except Exception as e:          # (the capture executes nothing: this very call goes on eagerly, the runner decides what the error means)
                self.capture_error = e
                self.ops = None
                torch.cuda.synchronize()
        if self.ops is None or not capture:          # (capture=False with a captured plan: this one step launch by launch - per-kernel event tables)
            grads = {}
            if self.arena is not None:
                # an eager pass in the arena: on the capture stream (see __init__), joined with the caller's stream on both sides
                cur, side = torch.cuda.current_stream(), _capture_stream(self.static['x'].device)
                side.wait_stream(cur)
                prev_scope, _hip.SCOPE = _hip.SCOPE, self.scope          # kernel scratch: the captured steps' own (one set, at its largest size once the arena is reserved)
                try:
                    with torch.cuda.stream(side), torch.cuda.use_mem_pool(self.arena):
                        result, launched = self._chain(None, grads)
                        self._finish(grads, launched)
                finally:
                    _hip.SCOPE = prev_scope
                cur.wait_stream(side)
                return result
            result, launched = self._chain(None, grads)
            self._finish(grads, launched)
            return result
        launched = 0
        for op in self.ops:
            if op[0] == 'graph':
                op[1].replay()
            elif op[0] == 'npos':
                self.dp._sum_small(op[1])
            else:
                self.dp.graph_launch(op[1], op[2])
                launched = op[2]
        self._finish(self.grads, launched)
        return self.result

    def _capture(self):
        """Record the launch sequence into hipGraph segments (nothing executes; the caller replays them right away)."""
        import gc
        cur = torch.cuda.current_stream()
        side = _capture_stream(self.static['x'].device)
        # Nothing may be destroyed while a capture is underway: a collected reference cycle that holds an older plan's CUDAGraph (or tensors of its
        # pool) releases graph memory from inside the capture and the runtime aborts the process.  Collect now, and keep the cyclic collector off
        # until the capture has ended (torch.cuda.graph() collects too, but leaves the collector on).
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        torch.cuda.synchronize()
        side.wait_stream(cur)
        # first with only the operand forms the last eager pass read (half of y2_prep_weights' traffic); should the algorithm table ask for
        # another one under capture (a shape it never measured), once more with all of them
        attempts = ([set(self.used_last)] if (PRUNE_OPERANDS and self.darknet and self.used_last and not _hip.split_mode()) else []) + [None]
        prev_scope = _hip.SCOPE
        aborted = []          # a pool handle dies with its last graph (torch asserts on reuse): the aborted attempt's graph lives until the next one exists
        try:
            for only in attempts:
                self.only = only
                if self.darknet and self.shared is not self.scope:
                    # shared operand buffers are ordinary allocations made BEFORE the capture: they outlive this plan's graphs and pool
                    _train_operands(self.dnn, self.static['x'].device, self.shared, only=only, alloc_only=True)
                seg = _Segments(self.pool, 'capture')
                grads = {}
                with torch.cuda.stream(side):
                    _hip.SCOPE = self.scope
                    try:
                        seg.begin()
                        self.result, _ = self._chain(seg, grads)
                        seg.end()
                        break
                    except OperandPruned:
                        aborted.append((seg, seg.graph))
                        seg.abort(_side_stream(self.static['x'].device))
                        if only is None:
                            raise
                    except BaseException:
                        seg.abort(_side_stream(self.static['x'].device))
                        raise
                    finally:
                        _hip.SCOPE = prev_scope
        finally:
            if gc_was_on:
                gc.enable()
        cur.wait_stream(side)
        self.ops, self.grads = seg.ops, grads
        del aborted[:]
        self._held = seg.held
        self._baked = self._addresses()

    def _addresses(self):
        """What a captured graph holds by ADDRESS besides its own pool: parameters, BatchNorm buffers, the wrapper's flat buckets."""
        a = [t.data_ptr() for t in self.params] + [t.data_ptr() for t in self.buffers]
        if self.dp is not None:
            a += [f.data_ptr() for f in self.dp._flat]
        return a

    def valid(self):
        """False when the memory a captured graph was recorded against has moved: `Train.eval` takes the model to the CPU and back between
        training steps (train.py:423-432) - the Parameters are the same objects afterwards, their storage is not."""
        return self.ops is None or self._baked == self._addresses()


_CAPTURE_STREAMS = {}


def _capture_stream(dev):
    s = _CAPTURE_STREAMS.get(str(dev))
    if s is None:
        s = _CAPTURE_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)      # (round 5, measured: a HIGH-priority capture stream for the main chain, so that the forked weight gradients only fill gaps: 35.0 instead of 30.7 ms per step)
    return s


_SCALE_DEV = {}


def _scale_tensor(sy, sx, dev):
    key = (sy, sx, str(dev))
    t = _SCALE_DEV.get(key)
    if t is None:
        t = _SCALE_DEV[key] = torch.tensor([sy, sx], dtype=torch.float32, device=dev).view(1, 1, 2)
    return t


def _hparam_tensor(w, dev):
    key = (w, str(dev))
    wd = _HPARAM_DEV.get(key)
    if wd is None:
        wd = _HPARAM_DEV[key] = torch.tensor(w, dtype=torch.float32, device=dev)
    return wd
