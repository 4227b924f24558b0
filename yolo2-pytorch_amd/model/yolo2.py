"""`model.yolo2` — the Darknet-19 YOLOv2 backbone plugin, MI355X-native.

Drop-in for the reference plugin model/yolo2.py:68-137: same constructor
`Darknet(config_channels, anchors, num_cls)`, same `state_dict()` keys and
shapes (weights stay in the reference's [Cout,Cin,k,k] layout so checkpoints,
`--finetune` and the darknet importer keep working), same
`forward(x[B,3,H,W]) -> [B, A*(5+C), H/32, W/32]`.  `[model] dnn =
model.yolo2.Darknet` in an unmodified config.ini selects this class.

Everything numerical runs in libyolo2_hip.so (include/yolo2_hip.h): the 23
convolutions are fp32-MFMA implicit GEMMs on NHWC activations with BatchNorm +
LeakyReLU folded into the epilogue; the five MaxPool2d(2), the reorg and the
concat (model/yolo2.py:79-130) are output addressing of the producing
convolution.  nn.Conv2d / nn.BatchNorm2d objects below are PARAMETER
CONTAINERS only (their forward is never called); the constructor is CPU-only
and lazy (no HIP call until a GPU tensor reaches forward), as required by the
reference's fork-after-init SummaryWorker (train.py:278-279).
"""
import ctypes

import torch
import torch.nn as nn

import model
import _hip

settings = {
    'size': (416, 416),
}

BN_EPS = 1e-5
LEAKY = 0.1


class Conv2d(nn.Module):
    """Parameter container with the reference's keys (model/yolo2.py:49-59): conv.weight [, conv.bias], bn.*"""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, stride=1, bn=True, act=True):
        nn.Module.__init__(self)
        if isinstance(padding, bool):
            padding = (kernel_size - 1) // 2 if padding else 0
        assert stride == 1 and padding == (kernel_size - 1) // 2, 'the HIP path implements stride-1 "same" convolutions'
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding=padding, bias=not bn)
        self.bn = nn.BatchNorm2d(out_channels, momentum=0.01) if bn else None
        self.has_act = act
        self.kernel_size = kernel_size

    def forward(self, x):
        raise RuntimeError('model.yolo2.Conv2d is a parameter container; the network runs through Darknet.forward (HIP)')


class _Pool(nn.Module):
    """Placeholder keeping the reference's nn.Sequential indices (MaxPool2d has no parameters)."""

    def forward(self, x):
        raise RuntimeError('pooling is fused into the producing convolution')


POOL = 'pool'
# Darknet-19 as data: per stage a list of POOL markers and (kernel, e, halved) entries whose width is (base << e) // (2 if halved
# else 1) with base = int(32 * ratio).  The order of the entries IS the order in which ConfigChannels is consulted (it is stateful:
# each call also sets the next layer's input width), and a layer's name is '<stage>.<position in the stage list>' - both are part
# of the checkpoint contract of the reference's plugin (model/yolo2.py:76-113).
_STAGE1 = [(3, 0, False), POOL, (3, 1, False), POOL,
           (3, 2, False), (1, 2, True), (3, 2, False), POOL,
           (3, 3, False), (1, 3, True), (3, 3, False), POOL,
           (3, 4, False), (1, 4, True), (3, 4, False), (1, 4, True), (3, 4, False)]
_STAGE2 = [POOL, (3, 5, False), (1, 5, True), (3, 5, False), (1, 5, True), (3, 5, False), (3, 5, False), (3, 5, False)]


def _build_stage(config_channels, prefix, spec, base, bn):
    mods = []
    for entry in spec:
        if entry is POOL:
            mods.append(_Pool())
            continue
        k, e, halved = entry
        width = (base << e) // (2 if halved else 1)
        cin = config_channels.channels
        cout = config_channels(width, '%s.%d.conv.weight' % (prefix, len(mods)))
        mods.append(Conv2d(cin, cout, k, bn=bn, padding=(k == 3)))
    return nn.Sequential(*mods)


def _init_reference(net, conv_init):
    """Initialisation of the reference plugins (model/yolo2.py:117-123, :165-171): conv weights by `conv_init`, BN gamma 1 / beta 0."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            conv_init(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class Darknet(nn.Module):
    def __init__(self, config_channels, anchors, num_cls, stride=2, ratio=1):
        nn.Module.__init__(self)
        if stride != 2:
            raise ValueError('the passthrough reorg is implemented for stride 2 (the only value the reference ships)')
        self.stride = stride
        bn = config_channels.config.getboolean('batch_norm', 'enable')
        base = int(32 * ratio)
        self.layers1 = _build_stage(config_channels, 'layers1', _STAGE1, base, bn)
        self.layers2 = _build_stage(config_channels, 'layers2', _STAGE2, base, bn)
        c_route = self.layers1[-1].conv.out_channels          # the 26x26 map both layers2 and the passthrough read
        c_deep = self.layers2[-1].conv.out_channels
        c_pass = config_channels(int(64 * ratio), 'passthrough.conv.weight')
        self.passthrough = Conv2d(c_route, c_pass, 1, bn=bn)
        c_cat = c_pass * stride * stride + c_deep             # reorg'ed passthrough channels first, then layers2 (model/yolo2.py:129)
        c_mix = config_channels(int(1024 * ratio), 'layers3.0.conv.weight')
        self.layers3 = nn.Sequential(Conv2d(c_cat, c_mix, 3, bn=bn, padding=True),
                                     Conv2d(c_mix, model.output_channels(len(anchors), num_cls), 1, bn=False, act=False))
        self.init()
        self._cache = None  # packed weights / folded BN for eval, keyed on parameter versions
        self._plans = _hip.PlanCache()  # execution plans (intermediate buffers + y2_conv_params arrays) per input shape, LRU
        self.grad_ready_hook = None  # set by train.DataParallelRCCL: called as hook(param, grad) inside backward, layer by layer
        self.profile = None  # tools: list receiving (kernel, flops, start_event, end_event) per conv launch

    def init(self):
        _init_reference(self, nn.init.kaiming_normal_)

    def scope(self, name):
        """Pruning helper (model/yolo2.py:132-133): 'layers1.4.conv.weight' -> 'layers1.4'."""
        return name.rsplit('.', 2)[0]

    def get_mapper(self, index):
        """Pruning helper (model/yolo2.py:135-137): how channel indices of the passthrough map onto the reorg'ed concat input."""
        if index != 94:
            return None
        copies = self.stride * self.stride
        return lambda indices, channels: torch.cat([indices + c * channels for c in range(copies)])

    # ------------------------------------------------------------------ execution plan
    def _blocks(self):
        """[(name, Conv2d, pool_follows)] in execution order for the three sequential stages."""
        def seq(prefix, s):
            out = []
            mods = list(s)
            for i, m in enumerate(mods):
                if isinstance(m, Conv2d):
                    out.append(('%s.%d' % (prefix, i), m, i + 1 < len(mods) and isinstance(mods[i + 1], _Pool)))
            return out
        return seq('layers1', self.layers1), seq('layers2', self.layers2), seq('layers3', self.layers3)

    def _first_block(self):
        return self.layers1[0]

    def backward_param_order(self):
        """Convolution weights in the order the training backward finishes their gradients (model.train_graph._darknet_bwd walks the
        forward's block list backwards: layers3, layers2, the passthrough branch, layers1): train.DataParallelRCCL buckets them in it."""
        b1, b2, b3 = self._blocks()
        fwd = [m for _, m, _ in b1] + [self.passthrough] + [m for _, m, _ in b2] + [m for _, m, _ in b3]
        return [m.conv.weight for m in reversed(fwd)]

    def _versions(self):
        """Cache key of everything derived from the parameters and buffers: (address, torch version counter) per tensor.  The
        raw-pointer writers (utils.optim, y2_bn_finalize) advance the counters of what they wrote through _hip.wrote, so the key is
        per tensor and per model: training another model does not touch it."""
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _weight_versions(self):
        """Cache key of what is derived from the convolution weights alone (the training step's GEMM operands): BatchNorm buffer
        updates of a forward pass do not move it."""
        return tuple((m.conv.weight.data_ptr(), m.conv.weight._version) for m in self.modules() if isinstance(m, Conv2d))

    @property
    def _plan_cache(self):
        """(tools / bench) the most recently used plan as (key, plan), None before the first forward."""
        plan = self._plans.latest()
        return None if plan is None else (plan['key'], plan)

    @_plan_cache.setter
    def _plan_cache(self, value):
        assert value is None
        self._plans.clear()

    def _prepare_eval(self, device):
        """Pack weights to [Cout][tap][Cin] and fold BN once per parameter version (y2_pack_weight / y2_bn_fold)."""
        ver = (device, self._versions(), _hip.split_mode())
        if self._cache is not None and self._cache[0] == ver:
            return self._cache[1]
        L = _hip.lib()
        st = _hip.stream()
        prep = {}
        first = self._first_block()
        for blk in [m for m in self.modules() if isinstance(m, Conv2d)]:
            w = blk.conv.weight.detach()
            _hip.require_gpu(w)
            cout, cin, k, _ = w.shape
            w = _hip.f32c(w)
            if blk is first:
                wp = w  # y2_conv0_fwd reads the state_dict layout directly
            else:
                wp = torch.empty(cout * cin * k * k, dtype=torch.float32, device=device)
                _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wp), cout, cin, k, 0, st), 'y2_pack_weight')
            if blk.bn is not None:
                scale = torch.empty(cout, dtype=torch.float32, device=device)
                shift = torch.empty(cout, dtype=torch.float32, device=device)
                bn = blk.bn
                _hip.check(L.y2_bn_fold(_hip.ptr(_hip.f32c(bn.weight.detach())), _hip.ptr(_hip.f32c(bn.bias.detach())),
                                        _hip.ptr(_hip.f32c(bn.running_mean)), _hip.ptr(_hip.f32c(bn.running_var)),
                                        BN_EPS, _hip.ptr(scale), _hip.ptr(shift), cout, st), 'y2_bn_fold')
            else:
                scale = None
                shift = _hip.f32c(blk.conv.bias.detach()) if blk.conv.bias is not None else None
            # Winograd-transformed copy of the deep 3x3 filters (the plan picks direct or Winograd per layer by measurement)
            u = _hip.wino_weight(wp, cout, cin) if (blk is not first and _hip.wino_eligible(cout, cin, k)) else None
            prep[blk] = (wp, scale, shift, u)
            if u is not None and _hip.SPLIT and cin % 32 == 0:
                prep.setdefault('split', {})[blk] = _hip.split_planes(u)      # bf16 plane triple of U (opt-in precision mode, Y2_ALGO_WINOGRAD_SPLIT)
        self._cache = (ver, prep)
        return prep

    def _conv_params(self, prep, blk, x, B, H, W, ldx, y=None, y_pool=None, ldy=0, coff=0, ldp=0, poff=0, out_mode=0):
        wp, scale, shift, _ = prep[blk]
        cout, cin = blk.conv.weight.shape[:2]
        p = _hip.ConvParams()
        p.x, p.w, p.scale, p.shift = x.data_ptr(), wp.data_ptr(), (scale.data_ptr() if scale is not None else None), (shift.data_ptr() if shift is not None else None)
        p.y = y.data_ptr() if y is not None else None
        p.y_pool = y_pool.data_ptr() if y_pool is not None else None
        p.stats = None
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, H, W, cin, ldx, cout, blk.kernel_size
        p.ldy, p.coff, p.ldp, p.poff, p.out_mode = ldy, coff, ldp, poff, out_mode
        p.slope = LEAKY if blk.has_act else 1.0
        p.tile = 0
        return p, 2.0 * cin * cout * blk.kernel_size ** 2 * B * H * W

    def _plan(self, prep, dev, B, cin0, H, W, slot=0):
        """Execution plan for one input shape: intermediate NHWC buffers (never exposed, reused across calls) and the
        y2_conv_params array of the 22 generic convolutions (model/yolo2.py:76-113 in execution order)."""
        # slot: plans of the same shape with PRIVATE intermediate buffers and scratch, so that two batches can be in flight on two streams
        # (detect.GraphedDetector(slot=...): the tail of one batch's kernels overlaps the head of the next batch's)
        # (the widths are part of the key: a plan's buffers, leading dimensions and algorithm choices are sized for the conv weights it was
        # built for - channel surgery, a replaced head - while a new parameter VERSION of the same shapes only moves operand pointers)
        widths = tuple(tuple(m.conv.weight.shape) for m in self.modules() if isinstance(m, Conv2d))
        key = (str(dev), B, cin0, H, W, _hip.tune_epoch(), _hip.WINOGRAD, _hip.FORCE_ALGO, _hip.split_mode(), slot, widths)
        plan = self._plans.get(key)
        if plan is not None:
            if plan['prep'] is not prep:
                # same shape, new parameter version (a training step ran in between): the buffers, algorithms and tiles stay, only the
                # operand pointers of the packed / folded / transformed weights move
                for p, blk in zip(plan['arr'], plan['blks']):
                    wp, scale, shift, u = prep[blk]
                    p.w = (prep['split'][blk] if p.algo in (4, 5) else u if p.algo in (1, 2, 3) else wp).data_ptr()
                    p.scale = scale.data_ptr() if scale is not None else None
                    p.shift = shift.data_ptr() if shift is not None else None
                plan['prep'] = prep
            return plan
        nbytes = [0]

        def new(*s):
            t = torch.empty(*s, dtype=torch.float32, device=dev)
            nbytes[0] += t.numel() * 4
            return t
        b1, b2, b3 = self._blocks()
        plist, flops, keep = [], 0.0, []

        ulist, blks = [], []

        def add(blk, *args, **kw):
            p, f = self._conv_params(prep, blk, *args, **kw)
            plist.append(p)
            ulist.append(prep[blk][3])
            blks.append(blk)
            return f
        name, blk0, pool = b1[0]
        c = blk0.conv.weight.shape[0]
        assert pool
        first = new(B, H // 2, W // 2, c)
        cur, h, w, ld = first, H // 2, W // 2, c
        full_last = None
        for i, (name, blk, pool) in enumerate(b1[1:], 1):
            c = blk.conv.weight.shape[0]
            if i == len(b1) - 1:
                # output feeds both the passthrough (full res) and layers2's leading MaxPool (model/yolo2.py:97,126-128)
                full_last = new(B, h, w, c)
                pooled = new(B, h // 2, w // 2, c)
                flops += add(blk, cur, B, h, w, ld, y=full_last, y_pool=pooled, ldy=c, ldp=c)
                cur, fh, fw = pooled, h, w
                h, w, ld = h // 2, w // 2, c
            elif pool:
                out = new(B, h // 2, w // 2, c)
                flops += add(blk, cur, B, h, w, ld, y_pool=out, ldp=c)
                cur, h, w, ld = out, h // 2, w // 2, c
            else:
                out = new(B, h, w, c)
                flops += add(blk, cur, B, h, w, ld, y=out, ldy=c)
                cur, ld = out, c
            keep.append(cur)
        # concat buffer [B, h, w, 4*c_pt + c_l2]; the passthrough writes its reorg'ed channels FIRST (model/yolo2.py:129)
        c_pt = self.passthrough.conv.weight.shape[0]
        c_l2 = b2[-1][1].conv.weight.shape[0]
        cat = new(B, h, w, 4 * c_pt + c_l2)
        flops += add(self.passthrough, full_last, B, fh, fw, full_last.shape[-1], y=cat, ldy=cat.shape[-1], coff=0, out_mode=1)
        for i, (name, blk, pool) in enumerate(b2):
            c = blk.conv.weight.shape[0]
            if i == len(b2) - 1:
                flops += add(blk, cur, B, h, w, ld, y=cat, ldy=cat.shape[-1], coff=4 * c_pt)
            else:
                out = new(B, h, w, c)
                flops += add(blk, cur, B, h, w, ld, y=out, ldy=c)
                cur, ld = out, c
                keep.append(out)
        cur, ld = cat, cat.shape[-1]
        head_index = None
        for i, (name, blk, pool) in enumerate(b3):
            c = blk.conv.weight.shape[0]
            if i == len(b3) - 1:
                head_index = len(plist)          # output buffer is allocated per call (it is returned to the caller)
                flops += add(blk, cur, B, h, w, ld, y=cur, ldy=c)
                head_shape = (B, h, w, c)
            else:
                out = new(B, h, w, c)
                flops += add(blk, cur, B, h, w, ld, y=out, ldy=c)
                cur, ld = out, c
                keep.append(out)
        for p, u, blk in zip(plist, ulist, blks):
            _hip.autotune_conv(p, dev, wino_w=u, wino_split=prep.get('split', {}).get(blk))      # per-layer algorithm + tile choice by measurement (cached per problem shape)
        need = max([_hip.lib().y2_conv_fwd_workspace_bytes(ctypes.byref(p)) for p in plist] + [0])
        if slot == 0:
            ws = _hip.workspace(dev, need) if need > 0 else None
        else:
            ws = new(int(need) // 4 + 4) if need > 0 else None      # (the per-device scratch is shared by everything that runs on ONE stream)
        for p in plist:
            p.workspace, p.workspace_bytes = (ws.data_ptr(), ws.numel() * 4) if ws is not None else (None, 0)
        arr = (_hip.ConvParams * len(plist))(*plist)
        # multiply-adds the MFMA pipe really executes: a Winograd layer runs 16 GEMMs over ceil(H/2)*ceil(W/2) tiles per image
        executed = sum(2.0 * p.Cin * p.Cout * (16 * p.B * ((p.H + 1) // 2) * ((p.W + 1) // 2) if p.algo in (1, 2, 3, 4, 5) else p.ksize ** 2 * p.B * p.H * p.W)
                       for p in plist)
        plan = dict(key=key, arr=arr, n=len(plist), first=first, head_index=head_index, head_shape=head_shape, flops=flops, flops_executed=executed,
                    algos=[int(p.algo != 0) for p in plist], blks=blks, prep=prep,
                    flops0=2.0 * cin0 * blk0.conv.weight.shape[0] * 9 * B * H * W, keep=(keep, full_last, cat, ws))
        self._plans.put(key, plan, nbytes[0])
        return plan

    def forward_nhwc(self, x, slot=0):
        """x [B,Cin,H,W] NCHW fp32 on the GPU -> head image [B, H/32, W/32, A*(5+C)] (NHWC, contiguous).
        Inference path (folded BatchNorm): one y2_conv0_fwd + one y2_conv_fwd_batch call.  `slot`: which private set of intermediate
        buffers to run in (calls that may overlap on different streams must use different slots)."""
        _hip.require_gpu(x)
        L = _hip.lib()
        x = _hip.f32c(x)
        B, cin0, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError('input size must be a multiple of 32 (got %dx%d)' % (H, W))
        dev = x.device
        prep = self._prepare_eval(dev)
        plan = self._plan(prep, dev, B, cin0, H, W, slot)
        st = _hip.stream()
        blk0 = self.layers1[0]
        wp, scale, shift, _ = prep[blk0]
        c = blk0.conv.weight.shape[0]
        out = torch.empty(plan['head_shape'], dtype=torch.float32, device=dev)
        plan['arr'][plan['head_index']].y = out.data_ptr()
        prof = self.profile
        if prof is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(wp), _hip.ptr(scale), _hip.ptr(shift), None, _hip.ptr(plan['first']), None,
                                  B, H, W, cin0, c, 0, c, LEAKY if blk0.has_act else 1.0, st), 'y2_conv0_fwd')
        if prof is not None:
            ev[1].record()
        _hip.check(L.y2_conv_fwd_batch(plan['arr'], plan['n'], st), 'y2_conv_fwd_batch')
        if prof is not None:
            ev[2].record()
            prof.append(('conv0', plan['flops0'], ev[0], ev[1]))
            prof.append(('conv_fwd', plan['flops'], ev[1], ev[2], plan['flops_executed'], sum(plan['algos'])))
        return out

    def forward(self, x):
        # BN semantics follow self.training alone, like nn.BatchNorm2d (model/yolo2.py:58): train() mode normalises with batch
        # statistics and updates the running ones whether or not autograd is recording (under no_grad the graph is simply
        # not taped); eval() mode runs the folded-BN inference chain - and stays differentiable like any nn.Module when autograd is
        # recording (frozen-BatchNorm fine-tuning; the reference back-propagates through `dnn` in eval mode,
        # receptive_field_analyzer.py:67,87): the backward then re-runs the layers with frozen statistics (train_graph).
        if self.training:
            from model import train_graph  # training graph (autograd.Function over the HIP kernels)
            return train_graph.darknet_forward(self, x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from model import train_graph
            return train_graph.darknet_forward_eval_grad(self, x)
        with torch.no_grad():
            out = self.forward_nhwc(x)
        # NCHW view of the NHWC head image: same values/shape as the reference's output; model.Inference's
        # permute(0,2,3,1).contiguous() (model/__init__.py:122) is then free.
        return out.permute(0, 3, 1, 2)


class _PadPool(nn.Module):
    """Placeholder for ConstantPad2d((0,1,0,1), float32.min) + MaxPool2d(2, stride=1) (model/yolo2.py:151-152): two entries
    in the reference's nn.Sequential, so two placeholders keep the state_dict indices."""

    def forward(self, x):
        raise RuntimeError('executed by y2_maxpool_fwd inside Tiny.forward_nhwc')


class Tiny(Darknet):
    """tiny-yolo (model/yolo2.py:140-173): 9 convolutions, state_dict keys `layers.{0,2,4,6,8,10,13,14}.{conv,bn}.*`,
    `layers.15.conv.{weight,bias}`; xavier_normal init.  Inference through the same kernels as Darknet (conv0 + LDS-DMA
    convolutions with fused pools); the stride-1 padded pool after layers.10 is one y2_maxpool_fwd launch."""

    def __init__(self, config_channels, anchors, num_cls, channels=16):
        nn.Module.__init__(self)
        bn = config_channels.config.getboolean('batch_norm', 'enable')
        # five conv+pool pairs (16..256), conv 512 + the padded stride-1 pool (two entries in the reference's Sequential), two conv 1024, head
        spec = []
        for e in range(5):
            spec += [(3, channels << e), _Pool]
        spec += [(3, channels << 5), _PadPool, _PadPool, (3, channels << 6), (3, channels << 6)]
        mods = []
        for entry in spec:
            if isinstance(entry, type):
                mods.append(entry())
                continue
            k, width = entry
            cin = config_channels.channels
            mods.append(Conv2d(cin, config_channels(width, 'layers.%d.conv.weight' % len(mods)), k, bn=bn, padding=True))
        mods.append(Conv2d(config_channels.channels, model.output_channels(len(anchors), num_cls), 1, bn=False, act=False))
        self.layers = nn.Sequential(*mods)
        self.init()
        self._cache = None
        self._plans = _hip.PlanCache()
        self.grad_ready_hook = None
        self.profile = None

    def init(self):
        _init_reference(self, nn.init.xavier_normal_)

    def _first_block(self):
        return self.layers[0]

    def backward_param_order(self):
        return [m.conv.weight for m in reversed([m for m in self.layers if isinstance(m, Conv2d)])]

    def forward_nhwc(self, x):
        _hip.require_gpu(x)
        L = _hip.lib()
        x = _hip.f32c(x)
        B, cin0, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError('input size must be a multiple of 32 (got %dx%d)' % (H, W))
        dev = x.device
        prep = self._prepare_eval(dev)
        st = _hip.stream()
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        mods = list(self.layers)
        blk0 = mods[0]
        wp, scale, shift, _ = prep[blk0]
        c = blk0.conv.weight.shape[0]
        cur = new(B, H // 2, W // 2, c)
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(wp), _hip.ptr(scale), _hip.ptr(shift), None, _hip.ptr(cur), None,
                                  B, H, W, cin0, c, 0, c, LEAKY if blk0.has_act else 1.0, st), 'y2_conv0_fwd')
        h, w, ld = H // 2, W // 2, c
        i = 2
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Conv2d):
                c = m.conv.weight.shape[0]
                pool = i + 1 < len(mods) and isinstance(mods[i + 1], _Pool)
                if pool:
                    out = new(B, h // 2, w // 2, c)
                    p, _ = self._conv_params(prep, m, cur, B, h, w, ld, y_pool=out, ldp=c)
                else:
                    out = new(B, h, w, c)
                    p, _ = self._conv_params(prep, m, cur, B, h, w, ld, y=out, ldy=c)
                _hip.autotune_conv(p, dev, wino_w=prep[m][3], wino_split=prep.get('split', {}).get(m))
                _hip.conv_workspace(p, dev)
                _hip.check(L.y2_conv_fwd(ctypes.byref(p), st), 'y2_conv_fwd')
                cur, ld = out, c
                if pool:
                    h, w = h // 2, w // 2
                    i += 1
            elif isinstance(m, _PadPool):
                out = new(B, h, w, ld)
                _hip.check(L.y2_maxpool_fwd(_hip.ptr(cur), _hip.ptr(out), B, h, w, ld, ld, ld, 2, 1, 0, 1, st), 'y2_maxpool_fwd')
                cur = out
                i += 1          # the pad + pool pair
            i += 1
        return cur

    def forward(self, x):
        if self.training:
            from model import train_graph
            return train_graph.tiny_forward(self, x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from model import train_graph
            return train_graph.tiny_forward(self, x, frozen=True)      # differentiable eval mode: the op-list graph with frozen statistics
        with torch.no_grad():
            out = self.forward_nhwc(x)
        return out.permute(0, 3, 1, 2)
