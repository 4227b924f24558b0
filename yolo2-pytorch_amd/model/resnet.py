"""`model.resnet` — the ResNet backbone plugins (resnet18 ... resnet152) of the reference, MI355X-native.

Drop-in for model/resnet.py:29-224: same constructors `resnetNN(config_channels, anchors, num_cls)`, same
`state_dict()` keys/shapes (`conv1.weight`, `bn1.*`, `layerL.B.conv{1,2,3}.weight`, `layerL.B.bn{1,2,3}.*`,
`layerL.B.downsample.{0,1}.*`, `conv.{weight,bias}`), same `forward(x[B,3,H,W]) -> [B, A*(5+C), H/32, W/32]`;
`[model] dnn = model.resnet.resnet50` in an unmodified ini selects it (BASELINE config 5 exercises the plugin swap).

Execution (inference): the NCHW input is converted once to zero-padded 4-channel NHWC (y2_nchw_to_nhwc); every
convolution — 7x7/s2 stem, 3x3/s2, 1x1/s2 down-sample, 1x1, 3x3 — is one y2_conv_fwd launch of the general fp32-MFMA
LDS-DMA kernel with BatchNorm folded into the epilogue, ReLU as LeakyReLU(slope 0), and the residual addition of
BasicBlock / Bottleneck (model/resnet.py:59,101) fused into the epilogue of the block's last convolution; the stem
max-pool is y2_maxpool_fwd.  The whole chain is one y2_conv_fwd_batch call per stage list, built once per input shape.
nn.Conv2d / nn.BatchNorm2d objects are parameter containers only.  Training runs through model/train_graph.py
(ResNetTrainFn: batch-statistics BN, strided data gradients as transposed convolutions, general weight gradient).
"""
import ctypes
import logging

import torch
import torch.nn as nn

import model
import _hip

BN_EPS = 1e-5


class _Block(nn.Module):
    """A residual block as data: SPEC lists its convolutions as (kernel, width multiplier, carries the block stride, padding).
    Attribute names conv<i>/bn<i>/downsample are the reference's state_dict keys (= torchvision's)."""
    SPEC = ()

    def __init__(self, config_channels, prefix, channels, stride=1):
        nn.Module.__init__(self)
        self.stride = stride
        c_in = config_channels.channels
        self._convs = []
        for i, (k, mult, strided, pad) in enumerate(self.SPEC, 1):
            cin = config_channels.channels
            cout = config_channels(channels * mult, '%s.conv%d.weight' % (prefix, i))
            s_i = stride if strided else 1
            conv, bn = nn.Conv2d(cin, cout, k, s_i, pad, bias=False), nn.BatchNorm2d(cout)
            setattr(self, 'conv%d' % i, conv)
            setattr(self, 'bn%d' % i, bn)
            self._convs.append((conv, bn, s_i, pad))
        c_out = config_channels.channels
        self.downsample = None
        if stride > 1 or c_in != c_out:      # projection shortcut (model/resnet.py:44-50, 86-92)
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride, bias=False), nn.BatchNorm2d(c_out))

    def convs(self):
        return list(self._convs)

    def forward(self, x):
        raise RuntimeError('model.resnet blocks are parameter containers; the network runs through ResNet.forward (HIP)')


class BasicBlock(_Block):
    """model/resnet.py:29-62: 3x3 (strided) -> 3x3."""
    SPEC = ((3, 1, True, 1), (3, 1, False, 1))


class Bottleneck(_Block):
    """model/resnet.py:65-104: 1x1 -> 3x3 (strided) -> 1x1 (4x wide)."""
    SPEC = ((1, 1, False, 0), (3, 1, True, 1), (1, 4, False, 0))


class ResNet(nn.Module):
    """model/resnet.py:107-159."""
    STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))       # (width, stride of the first block) of layer1..layer4

    def __init__(self, config_channels, anchors, num_cls, block, layers):
        nn.Module.__init__(self)
        c_in = config_channels.channels
        self.conv1 = nn.Conv2d(c_in, config_channels(64, 'conv1.weight'), kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(config_channels.channels)
        for li, ((width, stride), count) in enumerate(zip(self.STAGES, layers), 1):
            name = 'layer%d' % li
            blocks = [block(config_channels, '%s.%d' % (name, bi), width, stride if bi == 0 else 1) for bi in range(count)]
            setattr(self, name, nn.Sequential(*blocks))
        self.conv = nn.Conv2d(config_channels.channels, model.output_channels(len(anchors), num_cls), 1)
        for m in self.modules():       # model/resnet.py:120-126
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        self._cache = None
        self._plans = _hip.PlanCache()
        self.profile = None
        self.grad_ready_hook = None   # train.DataParallelRCCL: called as hook(param, grad) from inside backward

    def backward_param_order(self):
        """Convolution weights in the order the training backward finishes their gradients (the reverse of the forward's op list,
        model.train_graph.ResNetTrainFn: stem, per block the projection shortcut then its convolutions, head)."""
        fwd = [self.conv1]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                if blk.downsample is not None:
                    fwd.append(blk.downsample[0])
                fwd += [conv for conv, _, _, _ in blk.convs()]
        fwd.append(self.conv)
        return [c.weight for c in reversed(fwd)]

    def scope(self, name):
        """Pruning helper (model/resnet.py:142-156): the block-level scope of a parameter name - 'layer1.0.conv2.weight' ->
        'layer1.0.2', 'layer2.0.downsample.0.weight' -> 'layer2.0', 'conv.weight' -> 'conv'."""
        import re
        comp = name.split('.')[:-1]
        m = re.search(r'[(conv)|(bn)](\d+)', comp[-1])
        if m is not None:
            comp[-1] = m.group(1)
        elif len(comp) > 1:
            if comp[-2] != 'downsample':
                raise ValueError('unexpected parameter name %s' % name)
            comp = comp[:-1]
        elif comp[-1] != 'conv':
            raise ValueError('unexpected parameter name %s' % name)
        return '.'.join(comp)

    @property
    def _plan_cache(self):
        """(tools / bench) the most recently used plan as (key, plan), None before the first forward."""
        plan = self._plans.latest()
        return None if plan is None else (plan['key'], plan)

    @_plan_cache.setter
    def _plan_cache(self, value):
        assert value is None
        self._plans.clear()

    # ------------------------------------------------------------------ preparation: packed weights + folded BN
    def _versions(self):
        # torch version counters (raw-pointer writers - utils.optim, y2_bn_finalize - advance them through _hip.wrote)
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _prepare(self, dev):
        ver = (dev, self._versions())
        if self._cache is not None and self._cache[0] == ver:
            return self._cache[1]
        L = _hip.lib()
        st = _hip.stream()
        prep = {}

        wino = {}

        def fold(conv, bn):
            w = _hip.f32c(conv.weight.detach())
            _hip.require_gpu(w)
            cout, cin, k, _ = w.shape
            if cin % 4:                                   # the stem: zero-pad the input channels to the 4-channel NHWC image
                wpad = torch.zeros(cout, (cin + 3) // 4 * 4, k, k, dtype=torch.float32, device=dev)
                wpad[:, :cin] = w
                w, cin = wpad, wpad.shape[1]
            wp = torch.empty(w.numel(), dtype=torch.float32, device=dev)
            _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wp), cout, cin, k, 0, st), 'y2_pack_weight')
            if bn is not None:
                scale, shift = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
                _hip.check(L.y2_bn_fold(_hip.ptr(_hip.f32c(bn.weight.detach())), _hip.ptr(_hip.f32c(bn.bias.detach())), _hip.ptr(_hip.f32c(bn.running_mean)),
                                        _hip.ptr(_hip.f32c(bn.running_var)), BN_EPS, _hip.ptr(scale), _hip.ptr(shift), cout, st), 'y2_bn_fold')
            else:
                scale, shift = None, (_hip.f32c(conv.bias.detach()) if conv.bias is not None else None)
            prep[conv] = (wp, scale, shift, cin, cout, k)
            if _hip.wino_eligible(cout, cin, k, conv.stride[0]) and conv.padding[0] == 1:
                wino[id(wp)] = _hip.wino_weight(wp, cout, cin)     # Winograd copy of the stride-1 3x3 filters (conv2 of every block)
        fold(self.conv1, self.bn1)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                for conv, bn, _, _ in blk.convs():
                    fold(conv, bn)
                if blk.downsample is not None:
                    fold(blk.downsample[0], blk.downsample[1])
        fold(self.conv, None)
        prep['wino'] = wino
        self._cache = (ver, prep)
        return prep

    def _plan(self, prep, dev, B, cin0, H, W):
        widths = tuple(tuple(m.weight.shape) for m in self.modules() if isinstance(m, nn.Conv2d))      # (pruned / replaced layers re-plan)
        key = (str(dev), B, cin0, H, W, _hip.tune_epoch(), _hip.WINOGRAD, _hip.FORCE_ALGO, widths)
        plan = self._plans.get(key)
        if plan is not None:
            if plan['prep'] is not prep:      # same shape, new parameter version: only the weight operand pointers move
                for p, conv in zip([plan['stem']] + list(plan['arr']), plan['convs']):
                    wp, scale, shift, cin, cout, k = prep[conv]
                    u = prep['wino'].get(id(wp))
                    p.w = (u if p.algo in (1, 2, 3) else wp).data_ptr()
                    p.scale = scale.data_ptr() if scale is not None else None
                    p.shift = shift.data_ptr() if shift is not None else None
                plan['prep'] = prep
            return plan
        nbytes = [0]

        def new(*s):
            t = torch.empty(*s, dtype=torch.float32, device=dev)
            nbytes[0] += t.numel() * 4
            return t
        keep, flops, ulist, conv_order = [], [0.0], {}, []

        def conv_params(conv, x, h, w, ldx, y, stride, pad, slope, residual=None):
            wp, scale, shift, cin, cout, k = prep[conv]
            p = _hip.ConvParams()
            p.x, p.w = x.data_ptr(), wp.data_ptr()
            p.scale = scale.data_ptr() if scale is not None else None
            p.shift = shift.data_ptr() if shift is not None else None
            p.y, p.ldy = y.data_ptr(), cout
            p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, h, w, cin, ldx, cout, k
            p.stride, p.pad_plus1, p.slope, p.tile = stride, pad + 1, slope, 0
            ulist[id(p)] = prep['wino'].get(id(wp))
            conv_order.append(conv)
            if residual is not None:
                p.residual, p.ldr = residual.data_ptr(), cout
            ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
            flops[0] += 2.0 * conv.weight.shape[1] * cout * k * k * B * ho * wo
            return p

        cpad = (cin0 + 3) // 4 * 4
        x4 = new(B, H, W, cpad)
        c1 = self.conv1.weight.shape[0]
        h1, w1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        stem = new(B, h1, w1, c1)
        p_stem = conv_params(self.conv1, x4, H, W, cpad, stem, 2, 3, 0.0)
        h, w = (h1 + 2 - 3) // 2 + 1, (w1 + 2 - 3) // 2 + 1
        pooled = new(B, h, w, c1)
        plist = []
        cur, ld = pooled, c1
        keep += [x4, stem, pooled]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                s = blk.stride
                ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
                cout_blk = blk.convs()[-1][0].weight.shape[0]
                residual = cur
                if blk.downsample is not None:
                    residual = new(B, ho, wo, cout_blk)
                    plist.append(conv_params(blk.downsample[0], cur, h, w, ld, residual, s, 0, 1.0))   # BN folded, no activation
                    keep.append(residual)
                t, th, tw, tld = cur, h, w, ld
                convs = blk.convs()
                for i, (conv, bn, cs, cp) in enumerate(convs):
                    co = conv.weight.shape[0]
                    oh, ow = (th + 2 * cp - conv.kernel_size[0]) // cs + 1, (tw + 2 * cp - conv.kernel_size[0]) // cs + 1
                    out = new(B, oh, ow, co)
                    last = i == len(convs) - 1
                    plist.append(conv_params(conv, t, th, tw, tld, out, cs, cp, 0.0, residual=residual if last else None))   # ReLU = slope 0
                    keep.append(out)
                    t, th, tw, tld = out, oh, ow, co
                cur, h, w, ld = t, th, tw, tld
        head_index = len(plist)
        plist.append(conv_params(self.conv, cur, h, w, ld, cur, 1, 0, 1.0))
        head_shape = (B, h, w, self.conv.weight.shape[0])
        for p in [p_stem] + plist:
            _hip.autotune_conv(p, dev, wino_w=ulist.get(id(p))) if p is not plist[head_index] else None
        need = max([_hip.lib().y2_conv_fwd_workspace_bytes(ctypes.byref(p)) for p in [p_stem] + plist[:head_index]] + [0])
        ws = _hip.workspace(dev, need) if need > 0 else None
        for p in [p_stem] + plist:
            p.workspace, p.workspace_bytes = (ws.data_ptr(), ws.numel() * 4) if ws is not None else (None, 0)
        arr = (_hip.ConvParams * len(plist))(*plist)
        plan = dict(key=key, x4=x4, cpad=cpad, stem=p_stem, stem_out=stem, stem_hw=(h1, w1, c1), pooled=pooled, arr=arr, n=len(plist), head_index=head_index,
                    head_shape=head_shape, flops=flops[0], convs=conv_order, prep=prep, keep=(keep, ws))
        self._plans.put(key, plan, nbytes[0])
        return plan

    def forward_nhwc(self, x):
        _hip.require_gpu(x)
        L = _hip.lib()
        x = _hip.f32c(x)
        B, cin0, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError('input size must be a multiple of 32 (got %dx%d)' % (H, W))
        dev = x.device
        prep = self._prepare(dev)
        plan = self._plan(prep, dev, B, cin0, H, W)
        st = _hip.stream()
        out = torch.empty(plan['head_shape'], dtype=torch.float32, device=dev)
        plan['arr'][plan['head_index']].y = out.data_ptr()
        prof = self.profile
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _hip.check(L.y2_nchw_to_nhwc(_hip.ptr(x), _hip.ptr(plan['x4']), B, cin0, H, W, plan['cpad'], st), 'y2_nchw_to_nhwc')
        _hip.check(L.y2_conv_fwd(ctypes.byref(plan['stem']), st), 'y2_conv_fwd')
        h1, w1, c1 = plan['stem_hw']
        _hip.check(L.y2_maxpool_fwd(_hip.ptr(plan['stem_out']), _hip.ptr(plan['pooled']), B, h1, w1, c1, c1, c1, 3, 2, 1, 1, st), 'y2_maxpool_fwd')
        _hip.check(L.y2_conv_fwd_batch(plan['arr'], plan['n'], st), 'y2_conv_fwd_batch')
        if prof is not None:
            e1.record()
            prof.append(('conv_fwd', plan['flops'], e0, e1))
        return out

    def forward(self, x):
        if self.training:        # BN semantics follow self.training alone (see model.yolo2.Darknet.forward)
            from model import train_graph
            return train_graph.resnet_forward(self, x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from model import train_graph
            return train_graph.resnet_forward(self, x, frozen=True)     # differentiable eval mode: frozen BatchNorm statistics
        with torch.no_grad():
            out = self.forward_nhwc(x)
        return out.permute(0, 3, 1, 2)


def _make(block, layers):
    def ctor(config_channels, anchors, num_cls, **kwargs):
        net = ResNet(config_channels, anchors, num_cls, block, layers, **kwargs)
        try:
            pretrained = config_channels.config.getboolean('model', 'pretrained')
        except Exception:
            pretrained = False
        if pretrained:   # model/resnet.py:164-171 loads the torchvision model-zoo weights by URL
            raise RuntimeError('model.resnet: [model] pretrained=1 cannot be honoured (no torchvision model zoo / network here); '
                               'set pretrained=0 and load a checkpoint with load_state_dict (same keys as torchvision.models.resnet)')
        return net
    return ctor


resnet18 = _make(BasicBlock, [2, 2, 2, 2])
resnet34 = _make(BasicBlock, [3, 4, 6, 3])
resnet50 = _make(Bottleneck, [3, 4, 6, 3])
resnet101 = _make(Bottleneck, [3, 4, 23, 3])
resnet152 = _make(Bottleneck, [3, 8, 36, 3])
