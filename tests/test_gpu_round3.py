"""GPU tests of round 3: training at channel counts that are not multiples of 4 (pruned checkpoints, model/__init__.py:29-43),
differentiable eval() mode (frozen BatchNorm, receptive_field_analyzer.py:67,87), several train-mode forwards before a backward,
the launch-saving utility kernels (y2_multi, y2_small_dot / _scale, the step counter in y2_bn_finalize), the per-shape plan LRU
of the multi-scale schedule, and the fused weighted loss total of train.py:348-349."""
import configparser
import ctypes

import numpy as np
import pytest
import torch

from oracle import darknet as odark
from oracle import head as ohead
from oracle import loss as oloss
from oracle import synth
from oracle.make_golden import NARROW

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev():
    return torch.device('cuda:0')


def rel(got, ref):
    ref = ref.double()
    rms = ref.pow(2).mean().sqrt().item()
    return (got.double().cpu() - ref).abs().max().item() / max(rms, 1e-30)


def build(sd, num_cls=20, bn=True):
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1' if bn else '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, num_cls)
    dnn.load_state_dict(sd, strict=False)
    return model.Inference(cfg, dnn, anchors).to(dev()), anchors


def oracle_step(sd, x, data, anchors, training):
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    stats = {}
    f = odark.forward(x.double(), sd64, training=training, stats=stats if training else None)
    lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(f, anchors.double()), 0.6)
    oloss.total(lo).backward()
    return sd64, lo, stats, f


def check_grads(inf, sd64, tol=2e-3, floor=None):
    """Every parameter gradient against the fp64 oracle.  `floor`: the same gradients from the oracle run in fp32 - on an ill-conditioned
    draw (tiny maps, no BatchNorm: one LeakyReLU / max-pool decision that flips between fp32 and fp64 moves a small layer's gradient by
    percent) fp32 arithmetic itself is that far from fp64, and the bar is a small multiple of what torch's own fp32 does."""
    ours = dict(inf.dnn.named_parameters())
    worst = 0.0
    for k, v in sd64.items():
        if v.requires_grad:
            assert ours[k].grad is not None, k
            assert ours[k].grad.shape == v.grad.shape, k
            e = rel(ours[k].grad, v.grad)
            worst = max(worst, e)
            bar = tol if floor is None else max(tol, 3.0 * rel(floor[k].grad, v.grad))
            assert e <= bar, (k, e, bar)
    return worst


def oracle_step_fp32(sd, x, data, anchors, training):
    sd32 = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    f = odark.forward(x, sd32, training=training)
    lo, _ = oloss.loss(anchors, data, ohead.decode(f, anchors), 0.6)
    oloss.total(lo).backward()
    return sd32


UNALIGNED = [
    {'layers1.5': 6},                                               # the reference fixture's own width (oracle/make_golden.py: NARROW)
    {'layers1.0': 6, 'layers1.2': 10, 'layers1.5': 6, 'layers1.16': 30},     # first layer, a pooled layer and the route layer unaligned
    {'passthrough': 6, 'layers2.7': 62, 'layers3.0': 66},          # the concat buffer interleaves the padding of the reorg'ed copies
]


@pytest.mark.parametrize('widths', UNALIGNED, ids=['fixture', 'early', 'concat'])
@pytest.mark.parametrize('bn', [True, False])
def test_training_step_at_unaligned_channel_counts(widths, bn):
    """"kernels must accept arbitrary channel counts" - in training too.  Every gradient (in the parameters' own
    shapes), the loss terms and the running statistics against the oracle's fp64 autograd."""
    import model
    w = dict(NARROW)
    w['layers1.5'] = 8
    w.update(widths)
    # (seed 1 for the bias-only net at the 'early' widths: seed 0 draws a layers1.9 pre-activation of 1.2e-7 there, whose LeakyReLU branch
    # differs between fp32 and fp64 - one flipped element of a 432-pixel map moves that layer's weight gradient by 20 % of its rms;
    # a per-block tap (train_graph.DEBUG_TAP) showed every kernel of the block exact on its own inputs)
    sd = odark.init_state_dict(5, 20, seed=1 if (not bn and 'layers1.16' in widths and 'layers1.0' in widths) else 0, channels=w, head_scale=1 / 8.0, bn=bn)
    inf, anchors = build(sd, bn=bn)
    inf.train()
    S, B = 96, 3
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, 20, seed=2), S, S, S // 32, S // 32)
    pred = model._inference(inf, x.to(dev()))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    model.weighted_total(loss, oloss.HPARAM).backward()
    sd64, lo, stats, f = oracle_step(sd, x, data, anchors, True)
    assert rel(pred['feature'], f.detach()) <= 10 * TOL
    for k in lo:
        np.testing.assert_allclose(loss[k].item(), lo[k].item(), rtol=1e-4)
    check_grads(inf, sd64, floor=oracle_step_fp32(sd, x, data, anchors, True))
    bufs = dict(inf.dnn.named_buffers())
    for prefix, (rm, rv) in stats.items():
        np.testing.assert_allclose(bufs[prefix + '.bn.running_mean'].cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(bufs[prefix + '.bn.running_var'].cpu().numpy(), rv.numpy(), rtol=1e-4)
        assert int(bufs[prefix + '.bn.num_batches_tracked']) == 1


@pytest.mark.parametrize('widths', [{}, {'layers1.5': 6}], ids=['aligned', 'unaligned'])
def test_eval_mode_forward_is_differentiable_with_frozen_batchnorm(widths):
    """nn.Module semantics the reference relies on (receptive_field_analyzer.py:67,87; frozen-BN fine-tuning): eval() mode with autograd
    recording returns a graph; its gradients are those of the network with the RUNNING statistics, and nothing is updated."""
    import model
    w = dict(NARROW)
    w['layers1.5'] = 8
    w.update(widths)
    sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0)
    inf, anchors = build(sd)
    inf.eval()
    S, B = 96, 2
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, 20, seed=2), S, S, S // 32, S // 32)
    before = {k: v.clone() for k, v in inf.dnn.named_buffers()}
    pred = model._inference(inf, x.to(dev()))
    assert pred['feature'].requires_grad
    with torch.no_grad():
        plain = inf.dnn(x.to(dev()))
    assert torch.equal(plain, pred['feature'].detach())          # same inference chain, only taped
    loss, _ = model.loss(anchors, data, pred, 0.6)
    model.weighted_total(loss, oloss.HPARAM).backward()
    sd64, lo, _, f = oracle_step(sd, x, data, anchors, False)
    assert rel(pred['feature'], f.detach()) <= TOL
    for k in lo:
        np.testing.assert_allclose(loss[k].item(), lo[k].item(), rtol=1e-4)
    check_grads(inf, sd64, tol=2e-4)
    for k, v in inf.dnn.named_buffers():
        assert torch.equal(v, before[k]), k



def test_two_training_forwards_before_one_backward():
    """loss(model(a)) + loss(model(b)) - the second train-mode forward must not invalidate the first graph (its
    BatchNorm buffer updates are not weight changes); a no_grad forward in train() mode in between is harmless too; an optimizer step
    between forward and backward still raises."""
    import model
    import utils
    w = dict(NARROW)
    w['layers1.5'] = 8
    sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0)
    inf, anchors = build(sd)
    inf.train()
    S = 96
    xa, xb = synth.images(2, S, seed=1), synth.images(2, S, seed=5)
    da = synth.norm_data(synth.labels(2, S, 20, seed=2), S, S, 3, 3)
    db = synth.norm_data(synth.labels(2, S, 20, seed=6), S, S, 3, 3)

    def total(x, d):
        pred = model._inference(inf, x.to(dev()))
        l, _ = model.loss(anchors, d, pred, 0.6)
        return model.weighted_total(l, oloss.HPARAM)
    ta = total(xa, da)
    with torch.no_grad():
        inf.dnn(xa.to(dev()))
    tb = total(xb, db)
    (ta + tb).backward()
    both = {k: p.grad.clone() for k, p in inf.dnn.named_parameters()}
    # reference: the two gradients one at a time on a fresh copy (running statistics do not enter a train-mode gradient)
    inf2, _ = build(sd)
    inf2.train()
    acc = {}
    for x, d in ((xa, da), (xb, db)):
        for p in inf2.parameters():
            p.grad = None
        pred = model._inference(inf2, x.to(dev()))
        l, _ = model.loss(anchors, d, pred, 0.6)
        model.weighted_total(l, oloss.HPARAM).backward()
        for k, p in inf2.dnn.named_parameters():
            acc[k] = p.grad.clone() if k not in acc else acc[k] + p.grad
    for k in both:
        assert rel(both[k], acc[k].cpu()) <= 1e-4, k
    opt = utils.optim.SGD(inf.parameters(), lr=0.01)
    t = total(xa, da)
    opt.step()                                           # uses the gradients above; rewrites the weights through raw pointers
    with pytest.raises(RuntimeError, match='convolution weight was modified'):
        t.backward()


def test_multi_launch_and_small_kernels():
    import _hip
    d = dev()
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1000, generator=g).to(d)
    b = torch.randn(4097, generator=g).to(d)[1:]                  # 16-B alignment lost
    c64 = torch.randn(300, generator=g, dtype=torch.float64).to(d)
    i64 = torch.arange(10, device=d)
    dst = torch.empty(300, device=d)
    src = torch.randn(77, generator=g).to(d)
    cp = torch.empty(77, device=d)
    many = [torch.randn(5 + i, generator=g).to(d) for i in range(120)]      # more items than one table holds
    _hip.multi([(_hip.MULTI_ZERO, a, None), (_hip.MULTI_ZERO, b, None), (_hip.MULTI_F64_TO_F32, dst, c64), (_hip.MULTI_ZERO, i64, None),
                (_hip.MULTI_COPY, cp, src)] + [(_hip.MULTI_ZERO, t, None) for t in many])
    assert not a.any() and not b.any() and not i64.any() and all(not t.any() for t in many)
    assert torch.equal(dst, c64.float()) and torch.equal(cp, src)
    L = _hip.lib()
    v, w = torch.randn(5, generator=g).to(d), torch.tensor([5.0, 1, 1, 1, 1], device=d)
    out = torch.empty(1, device=d)
    _hip.check(L.y2_small_dot(_hip.ptr(v), _hip.ptr(w), 5, _hip.ptr(out), _hip.stream()), 'dot')
    want = 0.0
    for i in range(5):
        want = np.float32(want + np.float32(v[i].item()) * np.float32(w[i].item()))
    assert abs(out.item() - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
    sc = torch.empty(5, device=d)
    _hip.check(L.y2_small_scale(_hip.ptr(out), _hip.ptr(w), 5, _hip.ptr(sc), _hip.stream()), 'scale')
    assert torch.equal(sc, out * w)


def test_weighted_total_equals_the_reference_expression():
    import model
    A, C, rows, B = 5, 20, 3, 2
    gen = torch.Generator().manual_seed(9)
    feat = 0.5 * torch.randn(B, A * (5 + C), rows, rows, generator=gen)
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    data = synth.norm_data(synth.labels(B, 96, C, seed=3), 96, 96, rows, rows)

    class Id(torch.nn.Module):
        def forward(self, t):
            return t
    grads = []
    for fused in (True, False):
        f = feat.to(dev()).requires_grad_(True)
        l, dbg = model.loss(anchors, data, model._inference(model.Inference(None, Id(), anchors), f), 0.6)
        t = model.weighted_total(l, oloss.HPARAM) if fused else sum(l[k] * oloss.HPARAM[k] for k in l)
        assert t.shape == (1,)
        t.backward()
        grads.append((t.item(), f.grad.clone()))
        assert set(dbg.keys()) == {'iou', 'data', 'positive', 'negative'}
        assert dbg['positive'].dtype == torch.bool and dbg['negative'].shape == dbg['positive'].shape
        assert set(dbg['data'].keys()) == {'yx_min', 'yx_max', 'cls'}
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[1][0])
    assert rel(grads[0][1], grads[1][1].cpu()) <= 1e-6


def test_plan_cache_serves_the_multi_scale_schedule():
    """utils/data.py:135-141 changes the input size every few batches: the plans (buffers, algorithm choices) of the sizes seen stay
    cached, a revisit builds nothing, results are those of a fresh model, and a parameter update in between only moves operand pointers."""
    import _hip
    import utils
    import model
    sd = odark.init_state_dict(5, 20, seed=0, channels=NARROW, head_scale=1 / 8.0)
    inf, anchors = build(sd)
    inf.eval()
    xs = {S: synth.images(2, S, seed=S).to(dev()) for S in (64, 96, 128)}
    first = {}
    with torch.no_grad():
        for S in (64, 96, 128):
            first[S] = inf.dnn(xs[S]).clone()
        misses = inf.dnn._plans.misses
        for S in (96, 64, 128, 64):
            assert torch.equal(inf.dnn(xs[S]), first[S])
        assert inf.dnn._plans.misses == misses and len(inf.dnn._plans.d) == 3
        # a parameter update between visits: same plans, new weights
        with torch.no_grad():
            for p in inf.dnn.parameters():
                p.mul_(1.01)
        now = {k: v.detach().cpu().double() for k, v in inf.dnn.state_dict().items()}
        for S in (64, 128):
            got = inf.dnn(xs[S])
            assert rel(got, odark.forward(xs[S].cpu().double(), now)) <= TOL
        assert inf.dnn._plans.misses == misses
    small = _hip.PlanCache(entries=2)
    for k in range(4):
        small.put(k, {}, 10)
    assert list(small.d) == [2, 3]


@pytest.mark.parametrize('B,H,W,cin,cout,has_bn', [(2, 16, 32, 3, 32, 1), (1, 10, 16, 3, 40, 1), (3, 8, 16, 1, 8, 0), (2, 12, 48, 3, 64, 2), (2, 6, 32, 2, 6, 1)])
def test_first_layer_weight_gradient_without_a_materialised_dz(B, H, W, cin, cout, has_bn):
    """y2_conv0_wgrad_fused (BatchNorm / LeakyReLU / max-pool backward in the weight-gradient loader, from z, dy_pool and the pass-1 sums of
    y2_bn_act_bwd with dz = NULL) against the two-kernel form it replaces (y2_bn_act_bwd writing dz, then y2_conv0_wgrad)."""
    import _hip
    L, st, d = _hip.lib(), _hip.stream(), dev()
    g = torch.Generator().manual_seed(B + H + cout)
    x = torch.randn(B, cin, H, W, generator=g).to(d)
    z = torch.randn(B, H, W, cout, generator=g).to(d)
    z[:, ::2, 1::2] = z[:, ::2, ::2]                                  # ties inside pooling windows: the first maximum must win
    dyp = torch.randn(B, H // 2, W // 2, cout, generator=g).to(d)
    gamma = (torch.rand(cout, generator=g) + 0.5).to(d)
    mean, invstd = (torch.randn(cout, generator=g) * 0.1).to(d), (torch.rand(cout, generator=g) + 0.5).to(d)
    scale, shift = gamma * invstd, (torch.randn(cout, generator=g) * 0.1).to(d)
    args = lambda dz, sums: (_hip.ptr(z), _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(mean) if has_bn else None, _hip.ptr(invstd) if has_bn else None,
                             _hip.ptr(gamma) if has_bn else None, 0.1, None, 0, 0, 0, _hip.ptr(dyp), cout, 0, _hip.ptr(sums), _hip.ptr(dz), cout, B, H, W, cout, cout, has_bn, st)
    sums_a = torch.zeros(2 * cout, dtype=torch.float64, device=d)
    dz = torch.empty(B, H, W, cout, device=d)
    _hip.check(L.y2_bn_act_bwd(*args(dz, sums_a)), 'two-pass')
    want = torch.zeros(cout, cin, 3, 3, device=d)
    _hip.check(L.y2_conv0_wgrad(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(want), B, H, W, cin, cout, cout, st), 'conv0_wgrad')
    sums_b = torch.zeros(2 * cout, dtype=torch.float64, device=d)
    _hip.check(L.y2_bn_act_bwd(*args(None, sums_b)), 'sums only')
    assert rel(sums_b, sums_a.cpu()) <= 1e-5      # (fp32 block partials added in completion order)
    got = torch.zeros(cout, cin, 3, 3, device=d)
    _hip.check(L.y2_conv0_wgrad_fused(_hip.ptr(x), _hip.ptr(z), _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(mean) if has_bn else None, _hip.ptr(invstd) if has_bn else None,
                                      _hip.ptr(gamma) if has_bn else None, 0.1, _hip.ptr(dyp), cout, _hip.ptr(sums_b), _hip.ptr(got), B, H, W, cin, cout, cout, has_bn, st), 'fused')
    assert rel(got, want.cpu()) <= 2e-6, rel(got, want.cpu())


def test_detectors_in_different_slots_overlap_on_two_streams():
    """Serving-style pipelining: two captured detect steps with private intermediate buffers (slot 0 / 1) replayed concurrently on two HIP
    streams give exactly the results of serial replays (no shared scratch, tile counters or outputs)."""
    import detect
    sd = odark.init_state_dict(5, 20, seed=0, channels=NARROW, head_scale=1 / 8.0)
    inf, anchors = build(sd)
    inf.eval()
    xs = [synth.images(4, 160, seed=s).to(dev()) for s in (1, 2)]
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
    serial = []
    with torch.no_grad():
        for x in xs:
            d = detect.detect_batch(inf.dnn.forward_nhwc(x), anchors, **kw)
            serial.append({k: v.clone() for k, v in d.items()})
    runs = [detect.GraphedDetector(inf.dnn, anchors, x, static_input=True, slot=i, **kw) for i, x in enumerate(xs)]
    streams = [torch.cuda.Stream() for _ in runs]
    torch.cuda.synchronize()
    for rep in range(6):
        for r, s in zip(runs, streams):
            with torch.cuda.stream(s):
                r.run()
    torch.cuda.synchronize()
    for r, want in zip(runs, serial):
        for k in ('iou', 'yx_min', 'yx_max', 'prob', 'count', 'keep_count'):
            assert torch.equal(r.result[k], want[k]), k
        kc = want['keep_count'].tolist()
        for b, c in enumerate(kc):
            assert torch.equal(r.result['keep'][b, :c], want['keep'][b, :c])
    assert len(inf.dnn._plans.d) == 2          # one plan per slot


@pytest.mark.parametrize('arch,training', [('darknet', False), ('darknet', True), ('tiny', False), ('resnet18', False), ('resnet18', True)])
def test_gradient_with_respect_to_the_image(arch, training):
    """receptive_field_analyzer.py:67,87 back-propagates to the INPUT (in eval mode): d(feature element)/d(image) and every parameter gradient of
    the same backward against the oracle's autograd - Darknet-19, tiny-yolo and a ResNet, eval (frozen BatchNorm) and train mode."""
    import model
    import model.resnet
    import model.yolo2
    from oracle import resnet as ores
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}, 'model': {'pretrained': '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    if arch == 'darknet':
        w = dict(NARROW)
        w['layers1.5'] = 8
        sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0)
        dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, 20)
        fwd = lambda x, s: odark.forward(x, s, training=training)
    elif arch == 'tiny':
        sd = odark.init_tiny_state_dict(5, 20, seed=0, div=8, head_scale=0.25)
        dnn = model.yolo2.Tiny(model.ConfigChannels(cfg, sd), anchors, 20)
        fwd = lambda x, s: odark.tiny_forward(x, s, training=training)
    else:
        sd = ores.init_state_dict(arch, 5, 20, seed=0, width=8, head_scale=0.25)
        dnn = getattr(model.resnet, arch)(model.ConfigChannels(cfg, sd), anchors, 20)
        fwd = lambda x, s: ores.forward(x, s, arch, training=training)
    dnn.load_state_dict(sd, strict=False)
    dnn = dnn.to(dev())
    dnn.train(training)
    x = synth.images(2, 96, seed=1)
    xd = x.to(dev()).requires_grad_(True)
    f = dnn(xd)
    probe = torch.randn(f.shape, generator=torch.Generator().manual_seed(3))      # a fixed cotangent: sum(feature * probe)
    (f * probe.to(dev())).sum().backward()
    assert xd.grad is not None and xd.grad.shape == x.shape
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    (fwd(x64, sd64) * probe.double()).sum().backward()
    sd32 = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    x32 = x.clone().requires_grad_(True)
    (fwd(x32, sd32) * probe).sum().backward()
    floor = rel(x32.grad, x64.grad)
    assert rel(xd.grad, x64.grad) <= max(2e-4, 3 * floor), (rel(xd.grad, x64.grad), floor)
    ours = dict(dnn.named_parameters())
    for k, v in sd64.items():
        if v.requires_grad:
            assert rel(ours[k].grad, v.grad) <= max(2e-3, 3 * rel(sd32[k].grad, v.grad)), k


@pytest.mark.parametrize('cout,cin', [(40, 12), (100, 36), (64, 32), (256, 128), (33 * 4, 9 * 4)])
def test_multi_tensor_weight_preparation_equals_the_single_tensor_kernels(cout, cin):
    """y2_prep_weights (one launch for every layer's GEMM operands: packed, rotated / transposed for the data gradient, and the
    Winograd filter transforms of both) against y2_pack_weight + y2_wino_weight: bit-identical, for channel counts that do not fill
    its 32 x 8 staging tiles."""
    import _hip
    L, d = _hip.lib(), dev()
    g = torch.Generator().manual_seed(cout * 7 + cin)
    w3 = torch.randn(cout, cin, 3, 3, generator=g).to(d)
    w1 = torch.randn(cout, cin, 1, 1, generator=g).to(d)
    want = {}
    for name, w, k in (('3', w3, 3), ('1', w1, 1)):
        for mode in (0, 1):
            t = torch.empty(w.numel(), device=d)
            _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(t), cout, cin, k, mode, _hip.stream()), 'pack')
            want[(name, mode)] = t
    want[('3', 2)] = _hip.wino_weight(want[('3', 0)], cout, cin)
    want[('3', 3)] = _hip.wino_weight(want[('3', 1)], cin, cout)
    want[('3', 4)] = _hip.wino6_weight(want[('3', 1)], cin, cout)          # F(4x4,3x3) operand of the data gradient (Y2_PREP_WINO6_DGRAD, round 5)
    specs = [('3', 0, w3, 3, 1), ('3', 1, w3, 3, 1), ('1', 0, w1, 1, 1), ('1', 1, w1, 1, 1), ('3', 2, w3, 3, 16.0 / 9), ('3', 3, w3, 3, 16.0 / 9), ('3', 4, w3, 3, 4.0)]
    outs = [torch.full((int(round(w.numel() * f)),), float('nan'), device=d) for _, _, w, _, f in specs]
    table = (_hip.PrepItem * len(specs))()
    for i, (name, mode, w, k, f) in enumerate(specs):
        table[i].src, table[i].dst, table[i].Cout, table[i].Cin, table[i].ksize, table[i].mode = w.data_ptr(), outs[i].data_ptr(), cout, cin, k, mode
    _hip.check(L.y2_prep_weights(table, len(specs), _hip.stream()), 'y2_prep_weights')
    torch.cuda.synchronize()
    for (name, mode, w, k, f), got in zip(specs, outs):
        assert torch.equal(got, want[(name, mode)]), (name, mode)
