"""GPU parity tests of the training path: weight/data gradients, BN+LeakyReLU+pool forward/backward, region loss and a
full Darknet training step, against torch-CPU autograd in fp64 (oracle) and the reference-generated loss fixture."""
import configparser
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import darknet as odark
from oracle import head as ohead
from oracle import loss as oloss
from oracle import synth
from oracle.make_golden import NARROW

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev():
    return torch.device('cuda:0')


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel(got, ref):
    ref = ref.double()
    rms = ref.pow(2).mean().sqrt().item()
    return (got.double().cpu() - ref).abs().max().item() / max(rms, 1e-30)


@pytest.mark.parametrize('B,cin,cout,H,W,k', [(2, 32, 64, 8, 12, 3), (3, 64, 8, 13, 13, 1), (2, 4, 32, 16, 16, 3), (2, 64, 32, 16, 16, 3), (1, 32, 20, 9, 11, 3), (3, 256, 128, 13, 13, 3), (1, 128, 256, 26, 26, 3), (2, 40, 72, 7, 5, 3)])
def test_conv_wgrad_and_dgrad(B, cin, cout, H, W, k):
    import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, padding=(k - 1) // 2).backward(dz)
    d = dev()
    xd, dzd = nhwc(x.detach().float()).to(d), nhwc(dz.float()).to(d)
    dwp = torch.zeros(w.numel(), device=d)
    _hip.check(L.y2_conv_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, k, _hip.stream()), 'wgrad')
    dw = torch.empty(cout, cin, k, k, device=d)
    _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp), _hip.ptr(dw), cout, cin, k, _hip.stream()), 'unpack')
    assert rel(dw, w.grad) <= TOL
    # dgrad = forward kernel on rotated, in/out-swapped weights
    wd = torch.empty(w.numel(), device=d)
    wdev = w.detach().float().to(d).contiguous()
    _hip.check(L.y2_pack_weight(_hip.ptr(wdev), _hip.ptr(wd), cout, cin, k, 1, _hip.stream()), 'pack1')
    dx = torch.empty(B, H, W, cin, device=d)
    p = _hip.ConvParams()
    p.x, p.w, p.y = dzd.data_ptr(), wd.data_ptr(), dx.data_ptr()
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope = B, H, W, cout, cout, cin, k, cin, 1.0
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'dgrad')
    assert rel(dx.permute(0, 3, 1, 2), x.grad) <= TOL


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 32, 64, 8, 12), (3, 256, 128, 13, 13), (1, 128, 256, 26, 26), (2, 40, 72, 7, 5), (4, 512, 1024, 13, 13), (11, 16, 24, 13, 13), (9, 8, 12, 13, 11)])
def test_winograd_wgrad_and_dgrad(B, cin, cout, H, W):
    """y2_wino_wgrad (16 grouped reductions over tiles) and the Winograd data gradient (algo = 1 on the rotated filter)
    against fp64 autograd; odd sizes exercise the ragged last tile row / column."""
    import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, padding=1).backward(dz)
    d = dev()
    xd, dzd = nhwc(x.detach().float()).to(d), nhwc(dz.float()).to(d)
    need = L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout)
    ws = torch.empty(need // 4 + 4, device=d)
    dwp = torch.full((w.numel(),), 7.0, device=d)      # overwritten, not accumulated
    _hip.check(L.y2_wino_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, _hip.stream()), 'wino_wgrad')
    dw = torch.empty(cout, cin, 3, 3, device=d)
    _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp), _hip.ptr(dw), cout, cin, 3, _hip.stream()), 'unpack')
    assert rel(dw, w.grad) <= 4 * TOL
    wdev = w.detach().float().to(d).contiguous()
    # the same gradient from the transformed input a Winograd FORWARD leaves at the head of its workspace (no x needed)
    wp = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(wdev), _hip.ptr(wp), cout, cin, 3, 0, _hip.stream()), 'pack0')
    uf = _hip.wino_weight(wp, cout, cin)
    yf = torch.empty(B, H, W, cout, device=d)
    pf = _hip.ConvParams()
    pf.x, pf.w, pf.y, pf.algo = xd.data_ptr(), uf.data_ptr(), yf.data_ptr(), 1
    pf.B, pf.H, pf.W, pf.Cin, pf.ldx, pf.Cout, pf.ksize, pf.ldy, pf.slope = B, H, W, cin, cin, cout, 3, cout, 1.0
    wsf = torch.empty(L.y2_conv_fwd_workspace_bytes(ctypes.byref(pf)) // 4 + 4, device=d)
    pf.workspace, pf.workspace_bytes = wsf.data_ptr(), wsf.numel() * 4
    _hip.check(L.y2_conv_fwd(ctypes.byref(pf), _hip.stream()), 'wino fwd')
    dwp2 = torch.full((w.numel(),), -3.0, device=d)
    _hip.check(L.y2_wino_wgrad(None, _hip.ptr(dzd), _hip.ptr(dwp2), B, H, W, cin, cin, cout, cout, _hip.ptr(wsf), _hip.ptr(ws), ws.numel() * 4, _hip.stream()), 'wino_wgrad(v)')
    assert torch.equal(dwp2, dwp) or rel(dwp2, dwp.cpu()) <= 1e-5      # same arithmetic; split partial sums are added atomically
    # the 4x4-tile form (Winograd F(3x3, 4x4): larger transform constants - 1.2-1.4e-5 x rms in fp32), packed and native layouts
    for flags in (2, 3):
        dw6 = torch.full((w.numel(),), 5.0, device=d)
        _hip.check(L.y2_wino_wgrad_ex(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dw6), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, flags, _hip.stream()), 'wino6')
        if flags == 2:
            t6 = torch.empty(cout, cin, 3, 3, device=d)
            _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dw6), _hip.ptr(t6), cout, cin, 3, _hip.stream()), 'unpack')
        else:
            t6 = dw6.view(cout, cin, 3, 3)
        assert rel(t6, w.grad) <= 4 * TOL, (flags, rel(t6, w.grad))
    wd = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(wdev), _hip.ptr(wd), cout, cin, 3, 1, _hip.stream()), 'pack1')
    u = _hip.wino_weight(wd, cin, cout)
    dx = torch.empty(B, H, W, cin, device=d)
    p = _hip.ConvParams()
    p.x, p.w, p.y, p.algo = dzd.data_ptr(), u.data_ptr(), dx.data_ptr(), 1
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope = B, H, W, cout, cout, cin, 3, cin, 1.0
    _hip.conv_workspace(p, d)
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'wino dgrad')
    assert rel(dx.permute(0, 3, 1, 2), x.grad) <= 4 * TOL


def _random_wgrad_cases(n, seed=77):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 3]))
        out.append((int(rng.randint(1, 5)), int(rng.choice([4, 8, 20, 32, 64, 100, 128, 192])), int(rng.choice([4, 12, 32, 64, 100, 128, 256])),
                    int(rng.randint(1, 28)), int(rng.randint(1, 28)), k))
    return out


@pytest.mark.parametrize('B,cin,cout,H,W,k', _random_wgrad_cases(28))
def test_conv_wgrad_dgrad_random_shapes(B, cin, cout, H, W, k):
    """Seeded random shapes: direct weight gradient, Winograd weight gradient (3x3) and the data gradient (direct kernel on the
    rotated filter; Winograd where the library accepts it) against fp64 autograd."""
    import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(B * 131 + cin * 7 + cout * 3 + H + W + k)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, padding=(k - 1) // 2).backward(dz)
    d = dev()
    xd, dzd = nhwc(x.detach().float()).to(d), nhwc(dz.float()).to(d)
    dwp = torch.zeros(w.numel(), device=d)
    _hip.check(L.y2_conv_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, k, _hip.stream()), 'wgrad')
    dw = torch.empty(cout, cin, k, k, device=d)
    _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp), _hip.ptr(dw), cout, cin, k, _hip.stream()), 'unpack')
    assert rel(dw, w.grad) <= TOL
    if k == 3:
        ws = torch.empty(L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout) // 4 + 4, device=d)
        dwp2 = torch.full((w.numel(),), 9.0, device=d)
        _hip.check(L.y2_wino_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwp2), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, _hip.stream()), 'wino_wgrad')
        _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp2), _hip.ptr(dw), cout, cin, k, _hip.stream()), 'unpack')
        assert rel(dw, w.grad) <= 4 * TOL
    wd = torch.empty(w.numel(), device=d)
    wdev = w.detach().float().to(d).contiguous()
    _hip.check(L.y2_pack_weight(_hip.ptr(wdev), _hip.ptr(wd), cout, cin, k, 1, _hip.stream()), 'pack1')
    for algo in ((0, 1, 2, 3) if k == 3 else (0,)):
        dx = torch.full((B, H, W, cin), 3.0, device=d)
        p = _hip.ConvParams()
        src = wd if algo == 0 else _hip.wino_weight(wd, cin, cout)
        p.x, p.w, p.y, p.algo = dzd.data_ptr(), src.data_ptr(), dx.data_ptr(), algo
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope = B, H, W, cout, cout, cin, k, cin, 1.0
        if _hip.conv_workspace(p, d) < 0:
            continue            # e.g. the fused kernels want Cin % 32 == 0 (the implicit one Cin >= 64 too)
        _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'dgrad algo %d' % algo)
        assert rel(dx.permute(0, 3, 1, 2), x.grad) <= (4 if algo else 1) * TOL, algo


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 3, 32, 16, 32), (1, 3, 40, 9, 13), (2, 1, 8, 6, 6)])
def test_conv0_wgrad(B, cin, cout, H, W):
    import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(cout + W)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64)
    w = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    dz = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, padding=1).backward(dz)
    d = dev()
    dw = torch.zeros(cout, cin, 3, 3, device=d)
    xd, dzd = x.float().to(d), nhwc(dz.float()).to(d)
    _hip.check(L.y2_conv0_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dw), B, H, W, cin, cout, cout, _hip.stream()), 'conv0_wgrad')
    assert rel(dw, w.grad) <= TOL


@pytest.mark.parametrize('pool,both,C', [(False, False, 32), (True, False, 32), (True, True, 16), (False, False, 6), (True, True, 6)])
def test_bn_act_forward_backward(pool, both, C):
    import _hip
    L = _hip.lib()
    d = dev()
    B, H, W = 3, 8, 12
    g = torch.Generator().manual_seed(C + pool)
    z = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y = F.leaky_relu(F.batch_norm(z, rm, rv, gamma, beta, True, 0.01, 1e-5), 0.1)
    yp = F.max_pool2d(y, 2) if pool else None
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    dyp = torch.randn(yp.shape, generator=g, dtype=torch.float64) if pool else None
    lossv = (y * dy).sum() * (1.0 if (both or not pool) else 0.0)
    if pool:
        lossv = lossv + (yp * dyp).sum()
    lossv.backward()
    # ---- ours
    zd = nhwc(z.detach().float()).to(d)
    stats = torch.zeros(32, 2 * C, dtype=torch.float64)
    stats[5] = torch.stack([z.detach().sum((0, 2, 3)), (z.detach() ** 2).sum((0, 2, 3))]).reshape(-1)   # any of the Y2_STATS_REPL copies
    stats = stats.reshape(-1).to(d)
    scale, shift, mean, invstd = (torch.empty(C, device=d) for _ in range(4))
    rmd, rvd = torch.zeros(C, device=d), torch.ones(C, device=d)
    gd, bd = gamma.detach().float().to(d), beta.detach().float().to(d)
    n = B * H * W
    steps = torch.tensor(41, dtype=torch.int64, device=d)          # nn.BatchNorm2d.num_batches_tracked: incremented by the same launch
    _hip.check(L.y2_bn_finalize(_hip.ptr(stats), float(n), _hip.ptr(gd), _hip.ptr(bd), _hip.ptr(rmd), _hip.ptr(rvd), 0.01, 1e-5,
                                _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(mean), _hip.ptr(invstd), C, _hip.ptr(steps), _hip.stream()), 'fin')
    assert int(steps) == 42
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-5)
    yo = torch.empty(B, H, W, C, device=d)
    ypo = torch.empty(B, H // 2, W // 2, C, device=d) if pool else None
    _hip.check(L.y2_bn_act_fwd(_hip.ptr(zd), _hip.ptr(scale), _hip.ptr(shift), 0.1, _hip.ptr(yo), _hip.ptr(ypo), B, H, W, C, C, C, 0, C, 0, 0, _hip.stream()), 'fwd')
    assert rel(yo.permute(0, 3, 1, 2), y.detach()) <= TOL
    if pool:
        assert rel(ypo.permute(0, 3, 1, 2), yp.detach()) <= TOL
    sums = torch.zeros(2 * C, dtype=torch.float64, device=d)
    dz = torch.empty(B, H, W, C, device=d)
    dyf = nhwc(dy.float()).to(d) if (both or not pool) else None
    dypd = nhwc(dyp.float()).to(d) if pool else None
    _hip.check(L.y2_bn_act_bwd(_hip.ptr(zd), _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gd), 0.1,
                               _hip.ptr(dyf), C, 0, 0, _hip.ptr(dypd), C, 0, _hip.ptr(sums), _hip.ptr(dz), C, B, H, W, C, C, 1, _hip.stream()), 'bwd')
    assert rel(dz.permute(0, 3, 1, 2), z.grad) <= 5e-5
    np.testing.assert_allclose(sums[:C].cpu().numpy(), beta.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(sums[C:].cpu().numpy(), gamma.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_bn_act_reorg_roundtrip():
    """out_mode 1 (reorg into a concat buffer) forward, and fmode 1 gather in backward."""
    import _hip
    L = _hip.lib()
    d = dev()
    B, H, W, C = 2, 6, 10, 8
    z = torch.randn(B, C, H, W)
    zd = nhwc(z).to(d)
    cat = torch.full((B, H // 2, W // 2, 4 * C + 12), -3.0, device=d)
    _hip.check(L.y2_bn_act_fwd(_hip.ptr(zd), None, None, 0.1, _hip.ptr(cat), None, B, H, W, C, C, 4 * C + 12, 0, 0, 0, 1, _hip.stream()), 'fwd')
    ref = odark.reorg(F.leaky_relu(z, 0.1))
    assert torch.equal(cat[..., :4 * C].cpu().permute(0, 3, 1, 2), ref)
    assert torch.all(cat[..., 4 * C:] == -3.0)
    dcat = torch.randn(B, H // 2, W // 2, 4 * C + 12)
    sums = torch.zeros(2 * C, dtype=torch.float64, device=d)
    dz = torch.empty(B, H, W, C, device=d)
    dcd = dcat.to(d)
    _hip.check(L.y2_bn_act_bwd(_hip.ptr(zd), None, None, None, None, None, 0.1, _hip.ptr(dcd), 4 * C + 12, 0, 1, None, 0, 0,
                               _hip.ptr(sums), _hip.ptr(dz), C, B, H, W, C, C, 0, _hip.stream()), 'bwd')
    zz = z.clone().requires_grad_(True)
    (odark.reorg(F.leaky_relu(zz, 0.1)) * dcat[..., :4 * C].permute(0, 3, 1, 2)).sum().backward()
    np.testing.assert_allclose(dz.cpu().permute(0, 3, 1, 2).numpy(), zz.grad.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('onehot', [False, True])
def test_region_loss_matches_reference_fixture(golden, onehot):
    import model
    g = golden('loss')
    tag = 'onehot_' if onehot else 'ce_'
    gen = torch.Generator().manual_seed(11)
    feat = 0.5 * torch.randn(2, 125, 13, 13, generator=gen)
    f = feat.to(dev()).requires_grad_(True)
    anchors = torch.from_numpy(synth.ANCHORS_VOC)

    class Id(torch.nn.Module):
        def forward(self, t):
            return t
    inf = model.Inference(None, Id(), anchors)
    pred = model._inference(inf, f)
    data = synth.norm_data(synth.labels(2, 416, 20, seed=2, onehot=onehot), 416, 416, 13, 13)
    loss, debug = model.loss(anchors, data, pred, 0.6)
    total = sum(loss[k] * w for k, w in oloss.HPARAM.items())
    total.backward()
    for k in ('foreground', 'background', 'center', 'size', 'cls'):
        np.testing.assert_allclose(loss[k].item(), g[tag + k], rtol=2e-5)
    np.testing.assert_array_equal(debug['positive'].cpu().numpy(), g[tag + 'positive'].astype(bool))
    np.testing.assert_array_equal(debug['negative'].cpu().numpy(), g[tag + 'negative'].astype(bool))
    np.testing.assert_allclose(debug['iou'].cpu().numpy(), g[tag + 'best_iou'], rtol=1e-5, atol=1e-7)
    gr = g[tag + 'grad']
    np.testing.assert_allclose(f.grad.cpu().numpy(), gr, rtol=2e-4, atol=1e-6 * np.abs(gr).max())


@pytest.mark.parametrize('onehot', [False, True])
@pytest.mark.parametrize('case', synth.EDGE_CASES, ids=[c[0] for c in synth.EDGE_CASES])
def test_region_loss_edge_cases_match_reference_fixture(golden, case, onehot):
    """Image without objects, duplicate boxes, three boxes in one cell, first/last cell, 1-px and degenerate boxes, on the
    10/13/19 grids of the reference's training sizes: fixture from the reference (oracle/make_golden_loss_edge.py)."""
    import model
    name, S, rows = case
    g = golden('loss_edge')
    tag = '%s_%s_' % (name, 'onehot' if onehot else 'ce')
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    f = synth.edge_feature(5, 5, 20, rows, rows).to(dev()).requires_grad_(True)

    class Id(torch.nn.Module):
        def forward(self, t):
            return t
    pred = model._inference(model.Inference(None, Id(), anchors), f)
    data = synth.norm_data(synth.edge_labels(S, S, 20, onehot), S, S, rows, rows)
    loss, debug = model.loss(anchors, data, pred, 0.6)
    sum(loss[k] * w for k, w in oloss.HPARAM.items()).backward()
    for k in ('foreground', 'background', 'center', 'size', 'cls'):
        np.testing.assert_allclose(loss[k].item(), g[tag + k], rtol=3e-5)
    np.testing.assert_array_equal(debug['positive'].cpu().numpy().astype(bool), g[tag + 'positive'].astype(bool))
    np.testing.assert_array_equal(debug['negative'].cpu().numpy().astype(bool), g[tag + 'negative'].astype(bool))
    # IoUs of ~1e-2 against the 1-px box are differences of nearly equal fp32 coordinates: device expf vs host expf shows
    np.testing.assert_allclose(debug['iou'].cpu().numpy(), g[tag + 'best_iou'], rtol=1e-4, atol=1e-6)
    gr = g[tag + 'grad']
    np.testing.assert_allclose(f.grad.cpu().numpy(), gr, rtol=2e-4, atol=1e-6 * np.abs(gr).max())


def test_region_loss_coco_and_single_class_vs_oracle():
    import model
    for C, A in ((80, 5), (0, 5)):
        gen = torch.Generator().manual_seed(5 + C)
        ch = A * (5 + C)
        feat = 0.5 * torch.randn(3, ch, 10, 10, generator=gen)
        anchors = torch.from_numpy(synth.ANCHORS_VOC)
        data = synth.norm_data(synth.labels(3, 320, max(C, 1), seed=4), 320, 320, 10, 10)
        fo = feat.clone().double().requires_grad_(True)
        lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(fo, anchors.double()), 0.6)
        oloss.total(lo).backward()

        class Id(torch.nn.Module):
            def forward(self, t):
                return t
        f = feat.to(dev()).requires_grad_(True)
        pred = model._inference(model.Inference(None, Id(), anchors), f)
        l, _ = model.loss(anchors, data, pred, 0.6)
        sum(l[k] * oloss.HPARAM[k] for k in l).backward()
        assert set(l.keys()) == set(lo.keys())
        for k in l:
            np.testing.assert_allclose(l[k].item(), lo[k].item(), rtol=5e-5)
        gr = fo.grad.numpy()
        np.testing.assert_allclose(f.grad.cpu().numpy(), gr, rtol=5e-4, atol=2e-6 * np.abs(gr).max())


@pytest.mark.parametrize('rows,B,nmax,C,seed', [(10, 1, 1, 20, 1), (13, 3, 30, 20, 2), (19, 5, 8, 20, 3), (13, 2, 50, 80, 4), (10, 4, 3, 3, 5), (19, 1, 20, 0, 6)])
def test_region_loss_random_configurations_vs_oracle(rows, B, nmax, C, seed):
    """Grid sizes of the reference's multi-scale schedule, 1 ... 50 ground-truth boxes per image (more boxes than cells share
    anchors -> collisions), 0 / 3 / 20 / 80 classes: loss terms and d(loss)/d(feature) against the oracle's fp64 autograd."""
    import model
    A = 5
    S = rows * 32
    gen = torch.Generator().manual_seed(100 + seed)
    feat = 0.5 * torch.randn(B, A * (5 + C), rows, rows, generator=gen)
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    data = synth.norm_data(synth.labels(B, S, max(C, 1), nmax=nmax, seed=seed), S, S, rows, rows)
    fo = feat.clone().double().requires_grad_(True)
    lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(fo, anchors.double()), 0.6)
    oloss.total(lo).backward()

    class Id(torch.nn.Module):
        def forward(self, t):
            return t
    f = feat.to(dev()).requires_grad_(True)
    pred = model._inference(model.Inference(None, Id(), anchors), f)
    l, _ = model.loss(anchors, data, pred, 0.6)
    sum(l[k] * oloss.HPARAM[k] for k in l).backward()
    assert set(l.keys()) == set(lo.keys())
    for k in l:
        np.testing.assert_allclose(l[k].item(), lo[k].item(), rtol=5e-5)
    gr = fo.grad.numpy()
    np.testing.assert_allclose(f.grad.cpu().numpy(), gr, rtol=5e-4, atol=2e-6 * np.abs(gr).max())


GRAD_TOL = 2e-4        # README 'Tolerances': max |gradient error| / rms(gradient) against the oracle's fp64 autograd, or 2.5 x what the oracle's own fp32 run differs from it


def build(sd, num_cls=20, bn=True):
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1' if bn else '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, num_cls)
    dnn.load_state_dict(sd, strict=False)
    return model.Inference(cfg, dnn, anchors).to(dev()), anchors


@pytest.mark.parametrize('bn', [True, False])
def test_darknet_training_step_matches_oracle_autograd(bn):
    """fwd (batch-stat BN) + region loss + bwd on a narrow Darknet: every parameter gradient and the running statistics
    against the oracle's fp64 autograd (train.py:344-351 semantics)."""
    import model
    widths = dict(NARROW)
    widths['layers1.5'] = 8   # wgrad/dgrad DMA path needs channel counts that are multiples of 4
    sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0, bn=bn)
    inf, anchors = build(sd, bn=bn)
    inf.train()
    S, B = 96, 3
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, 20, seed=2), S, S, S // 32, S // 32)
    pred = model._inference(inf, x.to(dev()))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    total = sum(loss[k] * oloss.HPARAM[k] for k in loss)
    total.backward()
    # ---- oracle in fp64
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    stats = {}
    f = odark.forward(x.double(), sd64, training=True, stats=stats)
    lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(f, anchors.double()), 0.6)
    oloss.total(lo).backward()
    for k in lo:
        np.testing.assert_allclose(loss[k].item(), lo[k].item(), rtol=1e-4)
    # the same step on the oracle in fp32: where a gradient is a cancellation residue (the weights in front of a batch-statistics BatchNorm) fp32
    # itself is further from fp64 than 2e-4, whatever the implementation - the stated bound is max(2e-4, 2.5 x that floor)
    sd32 = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    l32, _ = oloss.loss(anchors, data, ohead.decode(odark.forward(x, sd32, training=True), anchors), 0.6)
    oloss.total(l32).backward()
    ours = dict(inf.dnn.named_parameters())
    worst = 0.0
    for k, v in sd64.items():
        if v.requires_grad:
            assert ours[k].grad is not None, k
            e = rel(ours[k].grad, v.grad)
            floor = rel(sd32[k].grad, v.grad)
            worst = max(worst, e)
            assert e <= max(GRAD_TOL, 2.5 * floor), (k, e, floor)
    bufs = dict(inf.dnn.named_buffers())
    for prefix, (rm, rv) in stats.items():
        np.testing.assert_allclose(bufs[prefix + '.bn.running_mean'].cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(bufs[prefix + '.bn.running_var'].cpu().numpy(), rv.numpy(), rtol=1e-4)
    print('worst relative gradient error', worst)


@pytest.mark.parametrize('bn', [False, True])
def test_eval_caches_follow_raw_pointer_writers(bn):
    """Packed / folded / Winograd-transformed weights are cached per parameter version, but the fused optimizer (and y2_bn_finalize)
    write parameter memory through raw pointers, invisibly to torch's version counters.  eval -> train step with utils.optim.SGD ->
    eval must see the NEW weights (with [batch_norm] enable=0 nothing else would invalidate the cache), and a captured
    GraphedDetector must refuse to replay its stale weights."""
    import detect
    import model
    import utils
    widths = dict(NARROW)
    widths['layers1.5'] = 8
    sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0, bn=bn)
    inf, anchors = build(sd, bn=bn)
    x = synth.images(2, 96, seed=1)
    data = synth.norm_data(synth.labels(2, 96, 20, seed=2), 96, 96, 3, 3)
    inf.eval()
    with torch.no_grad():
        f0 = inf.dnn(x.to(dev())).clone()
    graphed = detect.GraphedDetector(inf.dnn, anchors, x.to(dev()))
    graphed.run()
    inf.train()
    opt = utils.optim.SGD(inf.parameters(), lr=0.05, momentum=0.9)
    pred = model._inference(inf, x.to(dev()))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    opt.zero_grad()
    sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
    opt.step()
    inf.eval()
    with torch.no_grad():
        f1 = inf.dnn(x.to(dev())).clone()
        now = {k: v.detach().cpu().double() for k, v in inf.dnn.state_dict().items()}
        ref = odark.forward(x.double(), now)
    assert rel(f1, ref) <= TOL, rel(f1, ref)               # the updated weights (and running statistics) are what ran
    assert rel(f0, ref) > 100 * TOL                        # ... and they are not the old ones
    with pytest.raises(RuntimeError, match='parameters changed'):
        graphed.run()


# ------------------------------------------------------------------ ResNet training path (BASELINE config 5)
@pytest.mark.parametrize('B,cin,cout,H,W,k,stride,pad', [(2, 16, 32, 19, 19, 3, 2, 1), (2, 32, 64, 20, 20, 1, 2, 0), (1, 4, 64, 32, 40, 7, 2, 3), (2, 64, 64, 10, 10, 3, 2, 1), (2, 8, 16, 9, 9, 3, 1, 1)])
def test_strided_wgrad_and_transposed_dgrad(B, cin, cout, H, W, k, stride, pad):
    import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(cin + cout + H + k)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dz)
    Ho, Wo = y.shape[-2:]
    d = dev()
    xd, dzd = nhwc(x.detach().float()).to(d), nhwc(dz.float()).to(d)
    dwp = torch.zeros(w.numel(), device=d)
    _hip.check(L.y2_conv_wgrad_ex(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, k, stride, pad, _hip.stream()), 'wgrad_ex')
    dw = torch.empty(cout, cin, k, k, device=d)
    _hip.check(L.y2_unpack_weight_grad(_hip.ptr(dwp), _hip.ptr(dw), cout, cin, k, _hip.stream()), 'unpack')
    assert rel(dw, w.grad) <= TOL
    wd = torch.empty(w.numel(), device=d)
    wdev = w.detach().float().to(d).contiguous()
    _hip.check(L.y2_pack_weight(_hip.ptr(wdev), _hip.ptr(wd), cout, cin, k, 1, _hip.stream()), 'pack1')
    dx = torch.empty(B, H, W, cin, device=d)
    p = _hip.ConvParams()
    p.x, p.w, p.y = dzd.data_ptr(), wd.data_ptr(), dx.data_ptr()
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope = B, Ho, Wo, cout, cout, cin, k, cin, 1.0
    if stride == 1:
        p.stride, p.pad_plus1 = 1, k - 1 - pad + 1
    else:
        p.stride, p.pad_plus1, p.transposed, p.out_h, p.out_w = stride, pad + 1, 1, H, W
    _hip.conv_workspace(p, d)
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'dgrad')
    assert rel(dx.permute(0, 3, 1, 2), x.grad) <= TOL


def test_maxpool3x3s2_forward_backward():
    import _hip
    L = _hip.lib()
    d = dev()
    # (the 32-channel-multiple shapes take the LDS-tiled backward kernel: tiles of 8 x 32 pixels, ragged edges, several tiles per dimension, ties)
    for (B, C, H, W) in ((2, 8, 12, 14), (1, 6, 9, 9), (2, 32, 12, 14), (1, 64, 19, 23), (3, 32, 8, 32), (1, 32, 40, 70), (2, 64, 33, 65)):
        x = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
        if C == 32 and H == 40:
            x = (x * 2).round().div(2).detach().requires_grad_(True)          # many equal values: the FIRST maximum in scan order takes the gradient
        y = F.max_pool2d(x, 3, 2, 1)
        dy = torch.randn(y.shape, dtype=torch.float64)
        dy2 = torch.randn(y.shape, dtype=torch.float64)
        y.backward(dy + dy2)
        xd = nhwc(x.detach().float()).to(d)
        Ho, Wo = y.shape[-2:]
        yo = torch.empty(B, Ho, Wo, C, device=d)
        _hip.check(L.y2_maxpool_fwd(_hip.ptr(xd), _hip.ptr(yo), B, H, W, C, C, C, 3, 2, 1, 1, _hip.stream()), 'pool')
        assert torch.equal(yo.cpu().permute(0, 3, 1, 2), y.detach().float())
        dx = torch.empty(B, H, W, C, device=d)
        a, b = nhwc(dy.float()).to(d), nhwc(dy2.float()).to(d)
        _hip.check(L.y2_maxpool_bwd(_hip.ptr(xd), _hip.ptr(a), _hip.ptr(b), _hip.ptr(dx), B, H, W, C, C, C, C, 3, 2, 1, 1, _hip.stream()), 'pool_bwd')
        np.testing.assert_allclose(dx.cpu().permute(0, 3, 1, 2).numpy(), x.grad.float().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('arch', ['resnet18', 'resnet50'])
def test_resnet_training_step_matches_oracle_autograd(arch):
    import model
    import model.resnet
    from oracle import resnet as ores
    C = 20
    sd = ores.init_state_dict(arch, 5, C, seed=0, width=8, head_scale=0.25)
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}, 'model': {'pretrained': '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    net = getattr(model.resnet, arch)(model.ConfigChannels(cfg, sd), anchors, C)
    net.load_state_dict(sd, strict=False)
    inf = model.Inference(cfg, net, anchors).to(dev()).train()
    S, B = 96, 3
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, C, seed=2), S, S, S // 32, S // 32)
    pred = model._inference(inf, x.to(dev()))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    stats = {}
    f = ores.forward(x.double(), sd64, arch, training=True, stats=stats)
    lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(f, anchors.double()), 0.6)
    oloss.total(lo).backward()
    for k in lo:
        np.testing.assert_allclose(loss[k].item(), lo[k].item(), rtol=5e-4)   # fp32 vs fp64 through up to 53 batch-stat BN layers on 27 samples per channel
    # the same step on the oracle in fp32: its deviation from fp64 is the noise floor of fp32 arithmetic through ~50 batch-stat
    # BN layers, ReLU and max-pool decisions; the HIP path must stay within a small multiple of it
    sd32 = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    l32, _ = oloss.loss(anchors, data, ohead.decode(ores.forward(x, sd32, arch, training=True), anchors), 0.6)
    oloss.total(l32).backward()
    ours = dict(net.named_parameters())
    for k, v in sd64.items():
        if v.requires_grad:
            assert ours[k].grad is not None, k
            floor = rel(sd32[k].grad, v.grad)
            assert rel(ours[k].grad, v.grad) <= max(2e-3, 4 * floor), (k, rel(ours[k].grad, v.grad), floor)
    bufs = dict(net.named_buffers())
    for prefix, (rm, rv) in stats.items():
        np.testing.assert_allclose(bufs[prefix + '.running_mean'].cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(bufs[prefix + '.running_var'].cpu().numpy(), rv.numpy(), rtol=1e-4)


def test_tiny_training_step_matches_oracle_autograd():
    """model.yolo2.Tiny in training mode: conv (general kernel, 4-channel padded stem) + batch-stat BN (momentum
    0.01) + LeakyReLU, five MaxPool2d(2) and the padded stride-1 pool, region loss, full backward - against the oracle's
    fp64 autograd on the same seeded step."""
    import model
    import model.yolo2
    C = 20
    sd = odark.init_tiny_state_dict(5, C, seed=0, div=4, head_scale=0.25)
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    net = model.yolo2.Tiny(model.ConfigChannels(cfg, sd), anchors, C)
    assert not net.load_state_dict(sd, strict=False).unexpected_keys
    inf = model.Inference(cfg, net, anchors).to(dev()).train()
    S, B = 96, 3
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, C, seed=2), S, S, S // 32, S // 32)
    pred = model._inference(inf, x.to(dev()))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    stats = {}
    f = odark.tiny_forward(x.double(), sd64, training=True, stats=stats)
    lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(f, anchors.double()), 0.6)
    oloss.total(lo).backward()
    for k in lo:
        np.testing.assert_allclose(loss[k].item(), lo[k].item(), rtol=2e-4)
    sd32 = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    l32, _ = oloss.loss(anchors, data, ohead.decode(odark.tiny_forward(x, sd32, training=True), anchors), 0.6)
    oloss.total(l32).backward()
    ours = dict(net.named_parameters())
    for k, v in sd64.items():
        if v.requires_grad:
            assert ours[k].grad is not None, k
            floor = rel(sd32[k].grad, v.grad)      # fp32 noise floor of the same step on the oracle
            assert rel(ours[k].grad, v.grad) <= max(1e-3, 4 * floor), (k, rel(ours[k].grad, v.grad), floor)
    bufs = dict(net.named_buffers())
    for prefix, (rm, rv) in stats.items():
        np.testing.assert_allclose(bufs[prefix + '.bn.running_mean'].cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(bufs[prefix + '.bn.running_var'].cpu().numpy(), rv.numpy(), rtol=1e-4)


@pytest.mark.parametrize('kind', ['sgd', 'sgd_nesterov_wd', 'sgd_plain', 'adam', 'adam_wd'])
def test_fused_optimizer_matches_torch_optim(kind):
    """utils.optim (one launch per 48 tensors) against torch.optim on 60 tensors of assorted (odd, unaligned, large) sizes, 4 steps."""
    import utils
    d = dev()
    g = torch.Generator().manual_seed(3)
    shapes = [(125,), (1,), (3, 3, 3, 3), (1024, 1025), (7,), (64, 32, 3, 3), (33,), (4096,), (4097,), (2, 5)] * 6
    init = [torch.randn(*s, generator=g) for s in shapes]
    ours = [torch.nn.Parameter(t.clone().to(d)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(d)) for t in init]
    # an unaligned view: parameter storage offset by one float (16-B alignment lost -> scalar path)
    base_o, base_r = torch.randn(1001, generator=g).to(d), None
    base_r = base_o.clone()
    ours.append(torch.nn.Parameter(base_o[1:]))
    ref.append(torch.nn.Parameter(base_r[1:]))
    if kind.startswith('sgd'):
        kw = dict(sgd=dict(momentum=0.9), sgd_nesterov_wd=dict(momentum=0.8, nesterov=True, weight_decay=1e-2), sgd_plain=dict())[kind]
        a, b = utils.optim.SGD(ours, 0.05, **kw), torch.optim.SGD(ref, 0.05, **kw)
    else:
        kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=(1e-2 if kind == 'adam_wd' else 0.0))
        a, b = utils.optim.Adam(ours, 1e-2, **kw), torch.optim.Adam(ref, 1e-2, **kw)
    for step in range(4):
        for po, pr in zip(ours, ref):
            gr = torch.randn(po.shape, generator=g).to(d)
            po.grad, pr.grad = gr.clone(), gr.clone()
        if step == 2:       # a parameter that skips a step keeps its own step count / momentum state
            ours[3].grad = None
            ref[3].grad = None
        a.step()
        b.step()
    for po, pr in zip(ours, ref):
        np.testing.assert_allclose(po.detach().cpu().numpy(), pr.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa['state'].keys()) == set(sb['state'].keys())
    for k in sa['state']:
        assert set(sa['state'][k].keys()) == set(sb['state'][k].keys())


def test_fused_clip_grad_norm():
    import utils
    d = dev()
    g = torch.Generator().manual_seed(4)
    for scale, max_norm in ((1.0, 5.0), (1e-3, 5.0)):        # clipped / left alone
        ps = [torch.nn.Parameter(torch.zeros(*s, device=d)) for s in [(125,), (512, 513), (3,), (64, 64, 3, 3)] * 15]
        for p in ps:
            p.grad = (torch.randn(p.shape, generator=g) * scale).to(d)
        want = [p.grad.clone() for p in ps]
        tn = torch.nn.utils.clip_grad_norm_([torch.nn.Parameter(w) for w in want], max_norm) if False else None
        total = torch.sqrt(sum((w.double() ** 2).sum() for w in want))
        coef = min(1.0, max_norm / (total.item() + 1e-6))
        norm = utils.optim.clip_grad_norm_(ps, max_norm)
        np.testing.assert_allclose(norm.item(), total.item(), rtol=1e-6)
        for p, w in zip(ps, want):
            np.testing.assert_allclose(p.grad.cpu().numpy(), (w * coef).cpu().numpy(), rtol=2e-6, atol=1e-12)


def test_multi_scale_training_steps_with_fused_adam_and_clip():
    """train.py:338-362 over the reference's multi-scale schedule (config.ini sizes 320...608): consecutive steps at different input
    sizes through one model / optimizer (utils.optim.Adam, the ini's default optimizer, + gradient clipping): kernels take H, W at run
    time, per-shape tuning caches coexist, the loss stays finite and decreases on a repeated batch."""
    import train as y2train
    import utils
    widths = dict(NARROW)
    widths['layers1.5'] = 8   # training needs channel counts that are multiples of 4
    inf, anchors = build(odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0))
    inf.train()
    opt = utils.optim.Adam(inf.parameters(), 1e-3, betas=(0.9, 0.999), eps=1e-8)
    seen = {}
    for it, S in enumerate((320, 416, 352, 320, 416, 352, 320, 416, 352)):
        data = {k: v.to(dev()) for k, v in synth.labels(2, S, 20, seed=7).items()}
        data['tensor'] = synth.images(2, S, seed=8).to(dev())
        r = y2train.iterate(inf, opt, data, oloss.HPARAM, 0.6, anchors, clip=5.0)
        lt = float(r['loss_total'].detach())
        assert np.isfinite(lt), (it, S)
        seen.setdefault(S, []).append(lt)
    for S, ls in seen.items():
        assert ls[-1] < ls[0], (S, ls)              # the same batch at the same size: three Adam steps apart the loss went down


# ------------------------------------------------------------------ deterministic mode
def _one_training_step(S=96, B=3, seed=0):
    import model
    widths = dict(NARROW)
    widths['layers1.5'] = 8
    sd = odark.init_state_dict(5, 20, seed=seed, channels=widths, head_scale=1 / 8.0)
    inf, anchors = build(sd)
    inf.train()
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, 20, seed=2), S, S, S // 32, S // 32)
    pred = model._inference(inf, x.to(dev()))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
    torch.cuda.synchronize()
    out = {'loss.' + k: v.detach().cpu().clone() for k, v in loss.items()}
    out.update({'grad.' + k: p.grad.detach().cpu().clone() for k, p in inf.dnn.named_parameters()})
    out.update({'buf.' + k: b.detach().cpu().clone() for k, b in inf.dnn.named_buffers()})
    return out


def test_deterministic_mode_is_bit_reproducible_and_agrees_with_the_default_mode():
    """y2_set_deterministic: fixed-order reductions (split-K weight gradient, BatchNorm backward sums, BatchNorm statistics via
    y2_colstats_det, loss sums) and no timing-based algorithm selection.  Two runs of the same training step must agree BIT FOR BIT
    (losses, every parameter gradient, running statistics); the default (atomic) mode must agree with it to rounding."""
    import _hip
    base = _one_training_step()
    _hip.set_deterministic(True)
    try:
        assert _hip.lib().y2_get_deterministic() == 1
        a = _one_training_step()
        b = _one_training_step()
        c = _one_training_step()
    finally:
        _hip.set_deterministic(False)
    assert _hip.lib().y2_get_deterministic() == 0
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    worst = 0.0
    for k in a:
        if not a[k].is_floating_point():
            continue
        scale = base[k].double().abs().max().item()
        err = (a[k].double() - base[k].double()).abs().max().item() / max(scale, 1e-30)
        worst = max(worst, err)
        assert err <= 2e-3, (k, err)     # default-mode atomics and a possibly different (timed) algorithm choice: same tolerance as the fp64-oracle test
    print('deterministic vs default mode: worst relative difference %.2e' % worst)


def test_deterministic_wgrad_matches_fp64_and_repeats():
    """The K-split weight gradient under deterministic mode (partials + fixed-tree sum) against fp64 autograd, twice, bit-identical;
    direct, Winograd (grouped) and first-layer kernels."""
    import _hip
    L = _hip.lib()
    d = dev()
    _hip.set_deterministic(True)
    try:
        g = torch.Generator().manual_seed(5)
        B, cin, cout, H, W = 4, 64, 128, 26, 26
        x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        w = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
        dz = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
        F.conv2d(x, w, padding=1).backward(dz)
        xd, dzd = nhwc(x.detach().float()).to(d), nhwc(dz.float()).to(d)
        outs = []
        for rep in range(2):
            dwp = torch.full((w.numel(),), 3.0, device=d)          # deterministic mode writes, it does not accumulate
            _hip.check(L.y2_conv_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwp), B, H, W, cin, cin, cout, cout, 3, _hip.stream()), 'wgrad')
            ws = torch.empty(L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout) // 4 + 4, device=d)
            dwq = torch.full((w.numel(),), 3.0, device=d)
            _hip.check(L.y2_wino_wgrad(_hip.ptr(xd), _hip.ptr(dzd), _hip.ptr(dwq), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, _hip.stream()), 'wino_wgrad')
            outs.append((dwp.clone(), dwq.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for t, tol in ((outs[0][0], TOL), (outs[0][1], 4 * TOL)):
            dw = torch.empty(cout, cin, 3, 3, device=d)
            _hip.check(L.y2_unpack_weight_grad(_hip.ptr(t), _hip.ptr(dw), cout, cin, 3, _hip.stream()), 'unpack')
            assert rel(dw, w.grad) <= tol
        # first layer
        x0 = torch.randn(3, 3, 64, 64, generator=g, dtype=torch.float64, requires_grad=True)
        w0 = (torch.randn(32, 3, 3, 3, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
        dz0 = torch.randn(3, 32, 64, 64, generator=g, dtype=torch.float64)
        F.conv2d(x0, w0, padding=1).backward(dz0)
        x0d, dz0d = x0.detach().float().to(d).contiguous(), nhwc(dz0.float()).to(d)
        r = []
        for rep in range(2):
            dw0 = torch.full((32, 3, 3, 3), 5.0, device=d)
            _hip.check(L.y2_conv0_wgrad(_hip.ptr(x0d), _hip.ptr(dz0d), _hip.ptr(dw0), 3, 64, 64, 3, 32, 32, _hip.stream()), 'conv0_wgrad')
            r.append(dw0.clone())
        assert torch.equal(r[0], r[1]) and rel(r[0], w0.grad) <= TOL
        # the convolution refuses epilogue statistics in this mode (they would be atomics)
        p = _hip.ConvParams()
        st = torch.zeros(_hip.STATS_REPL * 2 * cout, dtype=torch.float64, device=d)
        y = torch.empty(B, H, W, cout, device=d)
        wp = torch.empty(w.numel(), device=d)
        p.x, p.w, p.y, p.stats = xd.data_ptr(), wp.data_ptr(), y.data_ptr(), st.data_ptr()
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope = B, H, W, cin, cin, cout, 3, cout, 1.0
        assert L.y2_conv_fwd(ctypes.byref(p), _hip.stream()) == -3
        # ... and y2_colstats_det gives the sums the epilogue would have
        z = torch.randn(B * H * W, cout, generator=g).to(d)
        _hip.colstats_det(z, B * H * W, cout, cout, st)
        ref = torch.cat([z.double().sum(0), (z.double() ** 2).sum(0)])
        assert torch.allclose(st[:2 * cout], ref, rtol=1e-12, atol=1e-9) and float(st[2 * cout:].abs().max()) == 0.0
    finally:
        _hip.set_deterministic(False)


@pytest.mark.parametrize('B,cin,cout,H,W,ldf,foff,has_bn', [(4, 16, 24, 13, 13, 24, 0, 1), (11, 8, 16, 13, 13, 32, 8, 1), (2, 32, 16, 26, 26, 16, 0, 1), (3, 12, 8, 8, 12, 8, 0, 2),
                                                            (1, 8, 8, 5, 7, 8, 0, 0), (9, 8, 12, 13, 11, 12, 0, 1)])
def test_bn_backward_fused_with_both_4x4_tile_transforms(B, cin, cout, H, W, ldf, foff, has_bn):
    """y2_bn_act_bwd_wino6 (BatchNorm / LeakyReLU backward pass 1 + pass 2 + both 4x4-tile Winograd transforms of dz in one kernel; autograd of
    model/yolo2.py:57-65 in front of a 3x3 convolution) against the forms it replaces: y2_bn_act_bwd (dz tensor) -> y2_conv_fwd F(4x4,3x3) on dz and
    y2_wino_wgrad_ex F(3x3,4x4) on (x, dz).  Same arithmetic, operation for operation; what differs is the order in which pass 1's fp32 partial sums meet
    (atomics), so everything agrees to 1e-5 x rms, not bit for bit.  Mosaic tile grids (11 and 9 images), channel windows in the gradient source, frozen statistics, no BN."""
    import _hip
    L, d = _hip.lib(), dev()
    g = torch.Generator().manual_seed(B * 100 + cin + cout + H)
    z = torch.randn(B, H, W, cout, generator=g).to(d)
    dy_buf = torch.randn(B, H, W, ldf, generator=g).to(d)
    x = torch.randn(B, H, W, cin, generator=g).to(d)
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(d), (torch.randn(cout, generator=g) * 0.1).to(d)
    mean = z.mean(dim=(0, 1, 2)).contiguous()
    invstd = torch.rsqrt(z.var(dim=(0, 1, 2), unbiased=False) + 1e-5).contiguous()
    scale = (gamma * invstd).contiguous() if has_bn else torch.ones(cout, device=d)
    shift = (beta - mean * scale).contiguous() if has_bn else beta
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(d).contiguous()          # forward weight [cout][cin]: its data gradient maps cout -> cin channels
    wd = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wd), cout, cin, 3, 1, _hip.stream()), 'pack1')
    u6 = _hip.wino6_weight(wd, cin, cout)
    args = (_hip.ptr(z), _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(mean) if has_bn else None, _hip.ptr(invstd) if has_bn else None, _hip.ptr(gamma) if has_bn else None, 0.1)
    # ---- three-kernel forms
    sums0 = torch.zeros(2 * cout, dtype=torch.float64, device=d)
    dz = torch.empty(B, H, W, cout, device=d)
    _hip.check(L.y2_bn_act_bwd(*args, _hip.ptr(dy_buf), ldf, foff, 0, None, 0, 0, _hip.ptr(sums0), _hip.ptr(dz), cout, B, H, W, cout, cout, has_bn, _hip.stream()), 'bn_act_bwd')

    def dgrad(src, algo, ldx):
        dx = torch.empty(B, H, W, cin, device=d)
        p = _hip.ConvParams()
        p.x, p.w, p.y = src.data_ptr(), u6.data_ptr(), dx.data_ptr()
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.slope, p.algo, p.tile = B, H, W, cout, ldx, cin, 3, cin, 1.0, algo, 0
        assert _hip.conv_workspace(p, d) >= 0
        _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'conv algo %d' % algo)
        return dx
    need = L.y2_wino_wgrad_workspace_bytes(B, H, W, cin, cout)
    ws = torch.empty(need // 4 + 4, device=d)

    def wgrad(src, flags):
        dw = torch.full((cout, cin, 3, 3), 5.0, device=d)
        _hip.check(L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(src), _hip.ptr(dw), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, flags, _hip.stream()), 'wino6 wgrad')
        return dw
    dx0, dw0 = dgrad(dz, 6, cout), wgrad(dz, 3)
    # ---- fused
    T = int(L.y2_wino6_tiles(B, H, W))
    assert 0 < T <= B * ((H + 3) // 4) * ((W + 3) // 4)
    sums1 = torch.zeros(2 * cout, dtype=torch.float64, device=d)
    v6, m6 = torch.full((36 * T * cout,), 7.0, device=d), torch.full((36 * T * cout,), 7.0, device=d)
    dz1 = torch.full((B, H, W, cout), 7.0, device=d)
    _hip.check(L.y2_bn_act_bwd_wino6(*args, _hip.ptr(dy_buf), ldf, foff, _hip.ptr(sums1), _hip.ptr(v6), _hip.ptr(m6), _hip.ptr(dz1), cout, B, H, W, cout, cout, has_bn, _hip.stream()), 'bn_act_bwd_wino6')
    dx1, dw1 = dgrad(v6, 7, cout), wgrad(m6, 7)
    torch.cuda.synchronize()
    # pass 1 is the same kernel; its fp32 partial sums meet in another order (LDS / fp64 atomics, parked per-workgroup partials), and dz depends on them
    assert rel(sums1, sums0.cpu()) <= 1e-5
    assert rel(dz1, dz.cpu()) <= 1e-5 and rel(dx1, dx0.cpu()) <= 1e-5 and rel(dw1, dw0.cpu()) <= 1e-5
    # only one of the two operands, and no plain gradient
    sums2 = torch.zeros(2 * cout, dtype=torch.float64, device=d)
    m6b = torch.empty(36 * T * cout, device=d)
    _hip.check(L.y2_bn_act_bwd_wino6(*args, _hip.ptr(dy_buf), ldf, foff, _hip.ptr(sums2), None, _hip.ptr(m6b), None, 0, B, H, W, cout, cout, has_bn, _hip.stream()), 'bn_act_bwd_wino6')
    assert rel(m6b, m6.cpu()) <= 1e-5
    assert L.y2_bn_act_bwd_wino6(*args, _hip.ptr(dy_buf), ldf, foff, _hip.ptr(sums2), None, None, None, 0, B, H, W, cout, cout, has_bn, _hip.stream()) != 0
    assert L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(m6), _hip.ptr(dw1), B, H, W, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, 5, _hip.stream()) != 0      # bit 2 without bit 1
