"""GPU tests of the opt-in split-bf16 precision mode (Y2_ALGO_WINOGRAD_SPLIT, csrc/gemm_split.hip): three bf16 planes per fp32
operand, six plane products per multiply on the bf16 matrix pipe.  The claim to verify is "fp32-level accuracy": every test holds
the mode to the tolerance of the fp32-MFMA path it replaces (the 3x3 convolutions of model/yolo2.py:76-113)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import ref_conv, rel_err, run_conv

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def test_split_planes_are_round_to_nearest_even_residuals():
    import _hip
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(4096, generator=g), torch.randn(4096, generator=g) * 1e-6, torch.randn(4096, generator=g) * 1e6,
                   torch.tensor([0.0, 1.0, -1.0, 3.0e38, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 + 3 * 2.0 ** -9, 2.0 ** -120])]).contiguous()
    planes = _hip.split_planes(x.to(dev())).view(3, -1).cpu()
    hi = x.to(torch.bfloat16)                       # torch converts round-to-nearest-even
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    assert torch.equal(planes[0].view(torch.int16), hi.view(torch.int16))
    assert torch.equal(planes[1].view(torch.int16), mid.view(torch.int16))
    assert torch.equal(planes[2].view(torch.int16), lo.view(torch.int16))
    back = planes[0].double() + planes[1].double() + planes[2].double()
    err = (back - x.double()).abs() / x.double().abs().clamp_min(1e-300)
    assert err[x != 0].max().item() <= 2.0 ** -23          # three 8-bit pieces carry fp32's 24 bits (up to the last rounding)


@pytest.mark.parametrize('M,N,K,groups', [(128, 128, 32, 1), (300, 200, 96, 3), (1568, 1024, 512, 2), (64, 64, 1280, 16), (129, 257, 64, 1), (5408, 512, 1024, 1)])
@pytest.mark.parametrize('bk', ['32x8', '32x4', '16x4'])
def test_gemm_split_matches_fp64(M, N, K, groups, bk, monkeypatch):
    """C = A B^T from plane triples against fp64, held to the error an exact-fp32 GEMM (fp32 products, fp32 accumulation) makes."""
    import _hip
    monkeypatch.setenv('Y2_SPLIT_BK', bk.split('x')[0])
    monkeypatch.setenv('Y2_SPLIT_WAVES', bk.split('x')[1])
    L = _hip.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(groups, M, K, generator=g)
    B = torch.randn(groups, N, K, generator=g) * 0.05
    A[:, :, ::7] *= 30.0                                     # mixed magnitudes inside a reduction
    Ad, Bd = A.to(dev()), B.to(dev())
    C = torch.full((groups, M, N + 3), -7.0, device=dev())   # row stride > N: the pad columns must stay untouched
    As, Bs = _hip.split_planes(Ad), _hip.split_planes(Bd)       # (kept alive: a temporary's memory would be recycled by the next allocation)
    _hip.check(L.y2_gemm_split(_hip.ptr(As), _hip.ptr(Bs), _hip.ptr(C), M, N, K, N + 3, groups, _hip.stream()), 'y2_gemm_split')
    want = torch.einsum('gmk,gnk->gmn', A.double(), B.double())
    f32 = torch.einsum('gmk,gnk->gmn', A, B)                 # CPU fp32 GEMM: the accuracy class claimed
    e_split, e_f32 = rel_err(C[:, :, :N].cpu(), want), rel_err(f32, want)
    print('gemm_split %dx%dx%d x%d: max|err|/rms %.3g (CPU fp32 GEMM %.3g)' % (M, N, K, groups, e_split, e_f32))
    assert e_split <= max(2.0 * e_f32, 1e-6), (e_split, e_f32)
    assert (C[:, :, N:] == -7.0).all()


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 64, 128, 12, 12), (3, 256, 128, 13, 13), (1, 128, 256, 26, 26), (2, 32, 64, 20, 20), (4, 512, 1024, 13, 13), (1, 1280, 64, 7, 9)])
@pytest.mark.parametrize('bk', ['32x8', '32x4', '16x4'])
def test_conv_split_mode_is_as_accurate_as_fp32_winograd(B, cin, cout, H, W, bk, monkeypatch):
    """y2_conv_fwd with algo = Y2_ALGO_WINOGRAD_SPLIT against fp64 F.conv2d (with BN affine + LeakyReLU, pooled output where the map
    is even): the tolerance of the fp32 Winograd tests (8e-5 x rms allowed, ~8e-6 measured) and within 1.5x of algo 1 on the same input."""
    monkeypatch.setenv('Y2_SPLIT_BK', bk.split('x')[0])
    monkeypatch.setenv('Y2_SPLIT_WAVES', bk.split('x')[1])
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    pool = H % 2 == 0 and W % 2 == 0
    z, want = ref_conv(x, w, scale, shift, 0.1, 3)
    errs = {}
    for algo in (1, 4, 5):
        out = run_conv(x, w, scale, shift, 0.1, 3, pool=pool, both=pool, wino=algo)
        errs[algo] = rel_err(out['y'].permute(0, 3, 1, 2), want)
        if pool:
            assert rel_err(out['y_pool'].permute(0, 3, 1, 2), F.max_pool2d(want, 2)) <= 8e-5
    print('conv %dx%d Cin %d Cout %d: split bf16x6 %.3g, split f16x3 %.3g, fp32 Winograd %.3g' % (H, W, cin, cout, errs[4], errs[5], errs[1]))
    for algo in (4, 5):
        assert errs[algo] <= 8e-5
        assert errs[algo] <= max(1.5 * errs[1], 4e-6), errs


def test_training_step_in_split_mode_matches_oracle_autograd(monkeypatch):
    """fprop and data gradients of the Winograd layers through Y2_ALGO_WINOGRAD_SPLIT (forced wherever the library accepts it; the filter
    transforms of all layers split by one y2_split_bf16x3 pass over their arena): losses, every parameter gradient and the running
    statistics against the oracle's fp64 autograd at the tolerance of the fp32 training test, and within 1e-4 of the fp32-MFMA run."""
    import _hip
    import model
    import test_gpu_round3 as R
    from oracle import darknet as odark, loss as oloss, synth
    from oracle.make_golden import NARROW
    w = dict(NARROW)
    w['layers1.5'] = 8
    sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0)
    S, B = 96, 3
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, 20, seed=2), S, S, S // 32, S // 32)
    grads = {}
    for mode in ('fp32', 'split', 'split16'):
        monkeypatch.setattr(_hip, 'SPLIT', {'fp32': '', 'split': 'bf16', 'split16': 'f16'}[mode])
        monkeypatch.setattr(_hip, 'FORCE_ALGO', 'winograd' if mode == 'fp32' else 'split')
        inf, anchors = R.build(sd)
        inf.train()
        calls = []
        if mode != 'fp32':
            L = _hip.lib()
            orig = L.y2_conv_fwd

            def spy(p, st):
                calls.append(p._obj.algo)
                return orig(p, st)
            monkeypatch.setattr(L, 'y2_conv_fwd', spy, raising=False)
        pred = model._inference(inf, x.to(dev()))
        loss, _ = model.loss(anchors, data, pred, 0.6)
        model.weighted_total(loss, oloss.HPARAM).backward()
        if mode != 'fp32':
            monkeypatch.undo()
            assert calls.count(4 if mode == 'split' else 5) >= (8 if mode == 'split' else 5), calls     # fprop (and, bf16 mode, dgrad) of the 64-channel 3x3 layers
        sd64, lo, stats, f = R.oracle_step(sd, x, data, anchors, True)
        for k in lo:
            np.testing.assert_allclose(loss[k].item(), lo[k].item(), rtol=1e-4)
        R.check_grads(inf, sd64)
        grads[mode] = {k: p.grad.detach().cpu() for k, p in inf.dnn.named_parameters()}
    for k in grads['fp32']:
        assert rel_err(grads['split'][k], grads['fp32'][k]) <= 1e-4, k
        assert rel_err(grads['split16'][k], grads['fp32'][k]) <= 1e-4, k


@pytest.mark.parametrize('M,N,K,groups', [(128, 128, 32, 1), (300, 200, 96, 3), (1568, 1024, 512, 2), (64, 64, 1280, 16), (5408, 512, 1024, 1)])
def test_gemm_split_f16_matches_fp64(M, N, K, groups):
    """The fp16 plane-pair GEMM with its power-of-two operand scales: values spread over six decades inside one reduction (what the fixed scales
    must absorb), against fp64 and the error of a CPU fp32 GEMM."""
    import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(groups, M, K, generator=g) * 4.0                # transformed activations: up to +-4 x the activation range
    A[:, :, ::7] *= 100.0
    A[:, :, 1::5] *= 1e-3
    B = torch.randn(groups, N, K, generator=g) * 0.03               # transformed weights
    B[:, ::3] *= 1e-2
    Ad, Bd = A.to(dev()), B.to(dev())
    As = torch.empty(2 * A.numel(), dtype=torch.float16, device=dev())
    Bs = torch.empty(2 * B.numel(), dtype=torch.float16, device=dev())
    _hip.check(L.y2_split_f16x2(_hip.ptr(Ad), _hip.ptr(As), A.numel(), 0.0625, _hip.stream()), 'split A')
    _hip.check(L.y2_split_f16x2(_hip.ptr(Bd), _hip.ptr(Bs), B.numel(), 256.0, _hip.stream()), 'split B')
    C = torch.full((groups, M, N), -7.0, device=dev())
    _hip.check(L.y2_gemm_split_f16(_hip.ptr(As), _hip.ptr(Bs), _hip.ptr(C), M, N, K, N, groups, 0.0625, _hip.stream()), 'y2_gemm_split_f16')
    want = torch.einsum('gmk,gnk->gmn', A.double(), B.double())
    f32 = torch.einsum('gmk,gnk->gmn', A, B)
    e_split, e_f32 = rel_err(C.cpu(), want), rel_err(f32, want)
    print('gemm_split_f16 %dx%dx%d x%d: max|err|/rms %.3g (CPU fp32 GEMM %.3g)' % (M, N, K, groups, e_split, e_f32))
    assert e_split <= max(2.0 * e_f32, 1e-6), (e_split, e_f32)


def test_f16_split_reports_operands_outside_its_range():
    """The fp16 plane mode has fp16's exponent range (times its fixed scales): an operand beyond it must be REPORTED (y2_split_f16_overflow),
    values inside it must not raise the flag."""
    import _hip
    _hip.split_overflowed()                                         # clear
    ok = torch.randn(4096).mul_(50.0).to(dev())
    _hip.split_planes(ok, 'f16')
    assert not _hip.split_overflowed()
    bad = ok.clone()
    bad[77] = 300.0                                                 # x 256 (the weight-operand scale) > 65504
    _hip.split_planes(bad, 'f16')
    assert _hip.split_overflowed()
    assert not _hip.split_overflowed()                              # reading cleared it
