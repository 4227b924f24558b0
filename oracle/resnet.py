"""torch-CPU restatement of the ResNet backbone plugin (model/resnet.py:29-159) — functional, state_dict in, feature out.

Follows: stem conv 7x7 s2 p3 -> BN -> ReLU -> MaxPool(3, s2, p1) (model/resnet.py:111-114,150-153); BasicBlock
(:29-62) / Bottleneck (:65-104) with the 1x1-stride down-sample branch when stride > 1 or the width changes (:39-45,
:78-84); head Conv2d(1x1, bias) (:118,158).  BatchNorm2d defaults: momentum 0.1, eps 1e-5.  The arithmetic of
conv/BN/pool lives in PyTorch (third-party dependency of the reference)."""
import collections
import math

import torch
import torch.nn.functional as F

from .darknet import output_channels

BN_EPS = 1e-5
LAYERS = {'resnet18': ('basic', [2, 2, 2, 2]), 'resnet34': ('basic', [3, 4, 6, 3]), 'resnet50': ('bottleneck', [3, 4, 6, 3]),
          'resnet101': ('bottleneck', [3, 4, 23, 3]), 'resnet152': ('bottleneck', [3, 8, 36, 3])}


def block_specs(arch, width=64):
    """[(prefix, kind, channels, stride)] in construction order (model/resnet.py:115-118,137-142)."""
    kind, layers = LAYERS[arch]
    out = []
    for li, (n, mult) in enumerate(zip(layers, (1, 2, 4, 8)), 1):
        for b in range(n):
            out.append(('layer%d.%d' % (li, b), kind, width * mult, (2 if li > 1 else 1) if b == 0 else 1))
    return out


def init_state_dict(arch='resnet50', num_anchors=5, num_cls=80, seed=0, width=64, randomize_bn=True, head_scale=1.0, dtype=torch.float32):
    """Synthetic weights with the reference's keys/shapes; width < 64 emulates a pruned checkpoint (ConfigChannels)."""
    g = torch.Generator().manual_seed(seed)
    sd = collections.OrderedDict()

    def conv(key, cout, cin, k):
        sd[key] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cin * k * k))

    def bn(prefix, c):
        if randomize_bn:
            sd[prefix + '.weight'] = torch.rand(c, generator=g) * 0.5 + 0.25
            sd[prefix + '.bias'] = torch.randn(c, generator=g) * 0.1
            sd[prefix + '.running_mean'] = torch.randn(c, generator=g) * 0.1
            sd[prefix + '.running_var'] = torch.rand(c, generator=g) + 0.5
        else:
            sd[prefix + '.weight'], sd[prefix + '.bias'] = torch.ones(c), torch.zeros(c)
            sd[prefix + '.running_mean'], sd[prefix + '.running_var'] = torch.zeros(c), torch.ones(c)

    conv('conv1.weight', width, 3, 7)
    bn('bn1', width)
    cin = width
    for prefix, kind, ch, stride in block_specs(arch, width):
        if kind == 'bottleneck':
            conv(prefix + '.conv1.weight', ch, cin, 1); bn(prefix + '.bn1', ch)
            conv(prefix + '.conv2.weight', ch, ch, 3); bn(prefix + '.bn2', ch)
            conv(prefix + '.conv3.weight', ch * 4, ch, 1); bn(prefix + '.bn3', ch * 4)
            cout = ch * 4
        else:
            conv(prefix + '.conv1.weight', ch, cin, 3); bn(prefix + '.bn1', ch)
            conv(prefix + '.conv2.weight', ch, ch, 3); bn(prefix + '.bn2', ch)
            cout = ch
        if stride > 1 or cin != cout:
            conv(prefix + '.downsample.0.weight', cout, cin, 1); bn(prefix + '.downsample.1', cout)
        cin = cout
    nout = output_channels(num_anchors, num_cls)
    sd['conv.weight'] = torch.randn(nout, cin, 1, 1, generator=g) * math.sqrt(2.0 / cin) * head_scale
    sd['conv.bias'] = torch.randn(nout, generator=g) * 0.1 * head_scale
    return collections.OrderedDict((k, v.to(dtype)) for k, v in sd.items())


def _bn(x, sd, prefix, training=False, stats=None):
    if training:
        rm, rv = sd[prefix + '.running_mean'].clone(), sd[prefix + '.running_var'].clone()
        y = F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'], True, 0.1, BN_EPS)
        if stats is not None:
            stats[prefix] = (rm, rv)
        return y
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.1, BN_EPS)


def forward(x, sd, arch='resnet50', training=False, stats=None):
    """model/resnet.py:149-158 (eval mode)."""
    width = sd['conv1.weight'].shape[0]
    x = F.relu(_bn(F.conv2d(x, sd['conv1.weight'], stride=2, padding=3), sd, 'bn1', training, stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for prefix, kind, ch, stride in block_specs(arch, width):
        residual = x
        if kind == 'bottleneck':
            out = F.relu(_bn(F.conv2d(x, sd[prefix + '.conv1.weight']), sd, prefix + '.bn1', training, stats))
            out = F.relu(_bn(F.conv2d(out, sd[prefix + '.conv2.weight'], stride=stride, padding=1), sd, prefix + '.bn2', training, stats))
            out = _bn(F.conv2d(out, sd[prefix + '.conv3.weight']), sd, prefix + '.bn3', training, stats)
        else:
            out = F.relu(_bn(F.conv2d(x, sd[prefix + '.conv1.weight'], stride=stride, padding=1), sd, prefix + '.bn1', training, stats))
            out = _bn(F.conv2d(out, sd[prefix + '.conv2.weight'], padding=1), sd, prefix + '.bn2', training, stats)
        if prefix + '.downsample.0.weight' in sd:
            residual = _bn(F.conv2d(x, sd[prefix + '.downsample.0.weight'], stride=stride), sd, prefix + '.downsample.1', training, stats)
        x = F.relu(out + residual)
    return F.conv2d(x, sd['conv.weight'], sd['conv.bias'])
