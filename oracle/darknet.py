"""torch-CPU restatement of the Darknet-19 YOLOv2 network (model/yolo2.py:33-130).

Functional (state_dict in, feature out) so the same code runs in fp32 and in
fp64 (ground truth for the conv tolerance, SURVEY.md 8d "Tolerance guidance").
The arithmetic of conv / batch-norm / max-pool lives in PyTorch (third-party,
pinned torch<=0.3.1 by requirements.txt:5); call sites restated here:
model/yolo2.py:57-59 (Conv2d = conv -> BN(momentum 0.01, eps 1e-5) -> LeakyReLU
0.1), :79,86,97 (MaxPool2d(2)), :33-46 (reorg), :125-130 (forward graph).
"""
import collections
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, model/yolo2.py:58
BN_MOMENTUM = 0.01  # model/yolo2.py:58
LEAKY = 0.1  # model/yolo2.py:59

# (prefix, kernel, default cout) in state_dict order; 'M' = MaxPool2d(2).  model/yolo2.py:76-113
LAYERS1 = [('layers1.0', 3, 32), 'M', ('layers1.2', 3, 64), 'M',
           ('layers1.4', 3, 128), ('layers1.5', 1, 64), ('layers1.6', 3, 128), 'M',
           ('layers1.8', 3, 256), ('layers1.9', 1, 128), ('layers1.10', 3, 256), 'M',
           ('layers1.12', 3, 512), ('layers1.13', 1, 256), ('layers1.14', 3, 512), ('layers1.15', 1, 256), ('layers1.16', 3, 512)]
LAYERS2 = ['M', ('layers2.1', 3, 1024), ('layers2.2', 1, 512), ('layers2.3', 3, 1024), ('layers2.4', 1, 512),
           ('layers2.5', 3, 1024), ('layers2.6', 3, 1024), ('layers2.7', 3, 1024)]
PASSTHROUGH = ('passthrough', 1, 64)
LAYERS3 = [('layers3.0', 3, 1024)]
HEAD = 'layers3.1'


def output_channels(num_anchors, num_cls):
    """model/__init__.py:46-50."""
    return num_anchors * (5 + num_cls) if num_cls > 1 else num_anchors * 5


def reorg(x, stride_h=2, stride_w=2):
    """model/yolo2.py:33-46: out[b,(dy*sw+dx)*C+c,i,j] = x[b,c,i*sh+dy,j*sw+dx]."""
    b, c, h, w = x.shape
    _h, _w = h // stride_h, w // stride_w
    x = x.view(b, c, _h, stride_h, _w, stride_w).permute(0, 3, 5, 1, 2, 4).contiguous()
    return x.view(b, stride_h * stride_w * c, _h, _w)


def init_state_dict(num_anchors=5, num_cls=20, seed=0, randomize_bn=True, head_scale=1.0, bn=True, channels=None, dtype=torch.float32):
    """Synthetic weights with the reference's keys/shapes:
    kaiming_normal(fan_in, gain sqrt2) conv weights (model/yolo2.py:117-120), BN gamma=1 beta=0 (:121-123),
    optionally randomised BN buffers so that folding is exercised.  `channels`
    (dict prefix->cout) overrides layer widths like a pruned checkpoint would
    (model/__init__.py:29-43)."""
    g = torch.Generator().manual_seed(seed)
    channels = channels or {}
    sd = collections.OrderedDict()

    def conv(prefix, cin, cout, k, with_bn):
        fan_in = cin * k * k
        sd[prefix + '.conv.weight'] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / fan_in)
        if with_bn:
            if randomize_bn:
                sd[prefix + '.bn.weight'] = torch.rand(cout, generator=g) + 0.5
                sd[prefix + '.bn.bias'] = torch.randn(cout, generator=g) * 0.1
                sd[prefix + '.bn.running_mean'] = torch.randn(cout, generator=g) * 0.1
                sd[prefix + '.bn.running_var'] = torch.rand(cout, generator=g) + 0.5
            else:
                sd[prefix + '.bn.weight'] = torch.ones(cout)
                sd[prefix + '.bn.bias'] = torch.zeros(cout)
                sd[prefix + '.bn.running_mean'] = torch.zeros(cout)
                sd[prefix + '.bn.running_var'] = torch.ones(cout)
        else:
            sd[prefix + '.conv.bias'] = torch.randn(cout, generator=g) * 0.1

    cin = 3
    for item in LAYERS1:
        if item == 'M':
            continue
        prefix, k, cout = item
        cout = channels.get(prefix, cout)
        conv(prefix, cin, cout, k, bn)
        cin = cout
    c_l1 = cin
    for item in LAYERS2:
        if item == 'M':
            continue
        prefix, k, cout = item
        cout = channels.get(prefix, cout)
        conv(prefix, cin, cout, k, bn)
        cin = cout
    c_l2 = cin
    prefix, k, cout = PASSTHROUGH
    c_pt = channels.get(prefix, cout)
    # state_dict order in the reference: layers1, layers2, passthrough, layers3 (model/yolo2.py:96,105,107,113)
    conv(prefix, c_l1, c_pt, k, bn)
    prefix, k, cout = LAYERS3[0]
    cout = channels.get(prefix, cout)
    conv(prefix, c_pt * 4 + c_l2, cout, k, bn)
    nout = output_channels(num_anchors, num_cls)
    fan_in = cout
    sd[HEAD + '.conv.weight'] = torch.randn(nout, cout, 1, 1, generator=g) * math.sqrt(2.0 / fan_in) * head_scale
    sd[HEAD + '.conv.bias'] = torch.randn(nout, generator=g) * 0.1 * head_scale
    return collections.OrderedDict((k, v.to(dtype)) for k, v in sd.items())


def conv_block(x, sd, prefix, k, training=False, stats=None):
    """model/yolo2.py:49-65 Conv2d.forward: conv (pad (k-1)//2, :52-56) -> bn -> leaky."""
    w = sd[prefix + '.conv.weight']
    bias = sd.get(prefix + '.conv.bias')
    x = F.conv2d(x, w, bias, stride=1, padding=(k - 1) // 2)
    if prefix + '.bn.weight' in sd:
        if training:
            rm = sd[prefix + '.bn.running_mean'].clone()
            rv = sd[prefix + '.bn.running_var'].clone()
            x = F.batch_norm(x, rm, rv, sd[prefix + '.bn.weight'], sd[prefix + '.bn.bias'], True, BN_MOMENTUM, BN_EPS)
            if stats is not None:
                stats[prefix] = (rm, rv)
        else:
            x = F.batch_norm(x, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'],
                             sd[prefix + '.bn.weight'], sd[prefix + '.bn.bias'], False, BN_MOMENTUM, BN_EPS)
    return F.leaky_relu(x, LEAKY)


def forward(x, sd, training=False, stats=None, taps=None):
    """model/yolo2.py:125-130.  x [B,3,H,W] -> feature [B,A(5+C),H/32,W/32].
    `taps` (dict) optionally receives every conv block's output (NCHW) by prefix."""
    def run(x, layers):
        for item in layers:
            if item == 'M':
                x = F.max_pool2d(x, 2)
            else:
                x = conv_block(x, sd, item[0], item[1], training, stats)
                if taps is not None:
                    taps[item[0]] = x
        return x
    x = run(x, LAYERS1)
    _x = conv_block(x, sd, PASSTHROUGH[0], 1, training, stats)
    if taps is not None:
        taps[PASSTHROUGH[0]] = _x
    _x = reorg(_x)
    x = run(x, LAYERS2)
    x = torch.cat([_x, x], 1)  # reorg channels FIRST, :129
    x = run(x, LAYERS3)
    x = F.conv2d(x, sd[HEAD + '.conv.weight'], sd[HEAD + '.conv.bias'])  # bn=False, act=False, :112
    if taps is not None:
        taps[HEAD] = x
    return x


# ------------------------------------------------------------------ tiny-yolo (model/yolo2.py:140-173)
TINY = [('layers.0', 16), 'M', ('layers.2', 32), 'M', ('layers.4', 64), 'M', ('layers.6', 128), 'M', ('layers.8', 256), 'M',
        ('layers.10', 512), 'P', ('layers.13', 1024), ('layers.14', 1024)]
TINY_HEAD = 'layers.15'


def init_tiny_state_dict(num_anchors=5, num_cls=20, seed=0, div=1, head_scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = collections.OrderedDict()
    cin = 3
    for item in TINY:
        if isinstance(item, str):
            continue
        prefix, cout = item[0], max(4, item[1] // div)
        sd[prefix + '.conv.weight'] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        sd[prefix + '.bn.weight'] = torch.rand(cout, generator=g) + 0.5
        sd[prefix + '.bn.bias'] = torch.randn(cout, generator=g) * 0.1
        sd[prefix + '.bn.running_mean'] = torch.randn(cout, generator=g) * 0.1
        sd[prefix + '.bn.running_var'] = torch.rand(cout, generator=g) + 0.5
        cin = cout
    nout = output_channels(num_anchors, num_cls)
    sd[TINY_HEAD + '.conv.weight'] = torch.randn(nout, cin, 1, 1, generator=g) * math.sqrt(2.0 / cin) * head_scale
    sd[TINY_HEAD + '.conv.bias'] = torch.randn(nout, generator=g) * 0.1 * head_scale
    return collections.OrderedDict((k, v.to(dtype)) for k, v in sd.items())


def tiny_forward(x, sd, training=False, stats=None):
    """model/yolo2.py:169-170: conv blocks, MaxPool2d(2), and ConstantPad2d((0,1,0,1), float32.min) + MaxPool2d(2, stride=1) (:151-152)."""
    for item in TINY:
        if item == 'M':
            x = F.max_pool2d(x, 2)
        elif item == 'P':
            x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), value=float(torch.finfo(torch.float32).min)), 2, stride=1)
        else:
            x = conv_block(x, sd, item[0], 3, training, stats)
    return F.conv2d(x, sd[TINY_HEAD + '.conv.weight'], sd[TINY_HEAD + '.conv.bias'])
