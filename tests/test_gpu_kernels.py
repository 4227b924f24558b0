"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, called through the C-ABI via the Python
mirror of the reference interface, against the CPU oracle / golden fixtures.

Tolerances (stated per SURVEY.md 8d): conv tensors max|err| <= 2e-5 * rms(reference) against an fp64 run of the
oracle; decode/softmax outputs 1e-5 relative (different exp implementation); IoU and NMS indices bit-exact.
"""
import configparser
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import darknet as odark
from oracle import detect as odet
from oracle import head as ohead
from oracle import iou as oiou
from oracle import nms as onms
from oracle import synth
from oracle.make_golden import NARROW

pytestmark = pytest.mark.gpu

CONV_TOL = 2e-5  # x rms of the fp64 reference


def dev():
    return torch.device('cuda:0')


def to_nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel_err(got, ref64):
    ref64 = ref64.double()
    rms = ref64.pow(2).mean().sqrt().item()
    return (got.double().cpu() - ref64).abs().max().item() / max(rms, 1e-30)


def run_conv(x_nchw, w, scale, shift, slope, k, pool=False, both=False, tile=0, reorg=False, coff=0, extra=0, stats=False, wino=False):
    """Drive y2_conv_fwd directly (NHWC in/out); returns dict of outputs as NCHW CPU tensors."""
    import _hip
    L = _hip.lib()
    d = dev()
    B, cin, H, W = x_nchw.shape
    cout = w.shape[0]
    x = to_nhwc(x_nchw).to(d)
    wd = w.to(d).contiguous()
    wp = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(wd), _hip.ptr(wp), cout, cin, k, 0, _hip.stream()), 'pack')
    p = _hip.ConvParams()
    sc = scale.to(d) if scale is not None else None
    sh = shift.to(d) if shift is not None else None
    if wino:
        u = torch.empty(16 * w.numel() // 9, device=d)
        _hip.check(L.y2_wino_weight(_hip.ptr(wp), _hip.ptr(u), cout, cin, _hip.stream()), 'wino_weight')
        wp, p.algo = u, int(wino)          # 1 = Winograd, 2 = Winograd with fused GEMM + output transform
        if int(wino) in (4, 5):            # ... 4 / 5 = three-kernel Winograd with the GEMMs on the bf16 / fp16 pipe (plane tuples)
            wp = _hip.split_planes(u, 'f16' if int(wino) == 5 else 'bf16')
    p.x, p.w = x.data_ptr(), wp.data_ptr()
    p.scale = sc.data_ptr() if sc is not None else None
    p.shift = sh.data_ptr() if sh is not None else None
    out = {}
    y = yp = st = None
    if reorg:
        ld = coff + 4 * cout + extra
        y = torch.full((B, H // 2, W // 2, ld), -7.0, device=d)
        p.y, p.ldy, p.coff, p.out_mode = y.data_ptr(), ld, coff, 1
    elif not pool or both:
        ld = coff + cout + extra
        y = torch.full((B, H, W, ld), -7.0, device=d)
        p.y, p.ldy, p.coff, p.out_mode = y.data_ptr(), ld, coff, 0
    if pool:
        ldp = cout + extra
        yp = torch.full((B, H // 2, W // 2, ldp), -7.0, device=d)
        p.y_pool, p.ldp, p.poff = yp.data_ptr(), ldp, 0
    if stats:
        st = torch.zeros(32 * 2 * cout, dtype=torch.float64, device=d)   # Y2_STATS_REPL copies
        p.stats = st.data_ptr()
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, H, W, cin, cin, cout, k
    p.slope, p.tile = slope, tile
    _hip.conv_workspace(p, d)      # split-K remainder scheme active whenever the library asks for scratch
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'conv')
    torch.cuda.synchronize()
    if y is not None:
        out['y'] = y.cpu()
    if yp is not None:
        out['y_pool'] = yp.cpu()
    if st is not None:
        out['stats'] = st.cpu().view(32, -1).sum(0)
    return out


def ref_conv(x, w, scale, shift, slope, k):
    z = F.conv2d(x.double(), w.double(), padding=(k - 1) // 2)
    u = z
    if scale is not None:
        u = u * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        u = u + shift.double().view(1, -1, 1, 1)
    return z, torch.where(u > 0, u, u * slope)


CASES = [
    # B, Cin, Cout, H, W, k, tile
    (2, 32, 64, 16, 24, 3, 1), (2, 32, 64, 16, 24, 3, 2), (2, 32, 64, 16, 24, 3, 3), (2, 32, 64, 16, 24, 3, 4), (2, 32, 64, 16, 24, 3, 5),
    (1, 64, 128, 13, 13, 3, 0), (2, 32, 64, 16, 24, 3, 7), (3, 512, 256, 13, 13, 3, 7), (2, 64, 125, 9, 7, 1, 7), (2, 16, 48, 10, 6, 3, 7), (3, 128, 64, 13, 13, 1, 0), (2, 96, 125, 7, 9, 1, 0), (2, 40, 72, 10, 6, 3, 1),
    (2, 64, 160, 20, 24, 3, 8), (2, 96, 288, 13, 13, 3, 9), (3, 128, 128, 16, 16, 1, 8), (2, 40, 72, 10, 6, 3, 9),
    # persistent workgroups (tile ids 11 / 12 / 13 / 15): more tiles than resident workgroups (several tiles per workgroup, odd and even slab counts per
    # tile: the LDS buffer parity carries over), fewer tiles than workgroups, ragged rows / channels, one K slab per tile, 3x3 taps
    (37, 64, 96, 40, 44, 1, 13), (37, 64, 96, 40, 44, 1, 15), (21, 96, 200, 52, 52, 1, 12), (9, 160, 136, 64, 60, 1, 11), (33, 32, 72, 48, 50, 1, 13),
    (2, 128, 64, 13, 13, 1, 15), (25, 32, 64, 36, 40, 3, 13), (12, 96, 125, 30, 34, 3, 15), (40, 224, 64, 40, 40, 1, 13),
    (2, 6, 20, 8, 8, 3, 0), (2, 64, 32, 16, 24, 3, 6), (3, 512, 256, 13, 13, 3, 5), (2, 1024, 125, 13, 13, 1, 3), (2, 64, 24, 16, 24, 3, 0), (1, 13, 33, 5, 7, 1, 2), (2, 256, 512, 13, 13, 3, 0),
]


@pytest.mark.parametrize('B,cin,cout,H,W,k,tile', CASES)
def test_conv_fwd_matches_fp64_reference(B, cin, cout, H, W, k, tile):
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + H + k)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    z, ref = ref_conv(x, w, scale, shift, 0.1, k)
    out = run_conv(x, w, scale, shift, 0.1, k, tile=tile, stats=True, extra=3, coff=2)
    y = out['y']
    assert torch.all(y[..., :2] == -7.0) and torch.all(y[..., 2 + cout:] == -7.0), 'wrote outside its channel window'
    assert rel_err(y[..., 2:2 + cout].permute(0, 3, 1, 2), ref) <= CONV_TOL
    s1, s2 = z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))
    np.testing.assert_allclose(out['stats'][:cout].numpy(), s1.numpy(), rtol=1e-5, atol=1e-5 * float(s2.max().sqrt()))
    np.testing.assert_allclose(out['stats'][cout:].numpy(), s2.numpy(), rtol=1e-5)


WINO_CASES = [
    # B, Cin, Cout, H, W, tile, pool, both
    (2, 32, 64, 16, 24, 0, False, False), (3, 512, 256, 13, 13, 0, False, False), (1, 64, 128, 13, 13, 3, False, False),
    (2, 128, 64, 7, 9, 5, False, False), (2, 256, 512, 26, 26, 1, False, False), (2, 32, 48, 12, 20, 2, True, False),
    (2, 64, 32, 8, 8, 0, True, True), (1, 16, 20, 5, 3, 0, False, False), (2, 1280, 1024, 13, 13, 0, False, False),
    (4, 32, 320, 64, 64, 0, False, False),      # 320 output tiles of 64x64 > 256 CUs: persistent workgroups of the fused kernel take 2 tiles
    (3, 64, 192, 50, 38, 0, True, True),        # ragged last row tile (1425 tiles of 2x2) x 3 channel tiles, pool + full output
]


@pytest.mark.parametrize('algo', [1, 2, 3])
@pytest.mark.parametrize('B,cin,cout,H,W,tile,pool,both', WINO_CASES)
def test_conv_fwd_winograd_matches_fp64_reference(B, cin, cout, H, W, tile, pool, both, algo):
    """algo = Y2_ALGO_WINOGRAD: F(2x2,3x3) input/filter/output transforms around the grouped MFMA GEMM.  Odd sizes (ragged
    last tile row/column), concat-style channel windows, negative scales before the fused pool.  The transforms cost a
    few ulps: tolerance 4x the direct kernel's (still ~1e-5 of the output rms against the fp64 truth)."""
    if algo >= 2 and cin % 32:
        pytest.skip('the fused kernel needs Cin % 32 == 0 (the library answers Y2_ENOSUP; autotune then keeps algo 1)')
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    scale = torch.randn(cout, generator=g)
    shift = torch.randn(cout, generator=g) * 0.1
    z, ref = ref_conv(x, w, scale, shift, 0.1, 3)
    out = run_conv(x, w, scale, shift, 0.1, 3, tile=tile, pool=pool, both=both, extra=0 if pool else 4, coff=0 if pool else 8, wino=algo, stats=True)
    s1, s2 = z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))      # training-mode BN statistics from the output transform (valid pixels only)
    np.testing.assert_allclose(out['stats'][:cout].numpy(), s1.numpy(), rtol=1e-5, atol=2e-5 * float(s2.max().sqrt()))
    np.testing.assert_allclose(out['stats'][cout:].numpy(), s2.numpy(), rtol=2e-5)
    if pool:
        assert rel_err(out['y_pool'].permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= 4 * CONV_TOL
    if both or not pool:
        y = out['y']
        c0 = 0 if pool else 8
        if not pool:
            assert torch.all(y[..., :8] == -7.0) and torch.all(y[..., 8 + cout:] == -7.0), 'wrote outside its channel window'
        assert rel_err(y[..., c0:c0 + cout].permute(0, 3, 1, 2), ref) <= 4 * CONV_TOL


@pytest.mark.parametrize('B,cin,cout,H,W,pad_ch,pool', [
    (2, 64, 64, 13, 13, 0, False),        # ragged last tile row / column, 98 tiles: two tile rows of 64, the second mostly empty
    (3, 64, 96, 26, 26, 32, False),       # input is a channel window of a wider tensor (ldx > Cin), Cout not a multiple of 64
    (1, 128, 64, 52, 52, 0, True),        # 676 tiles: several tiles per workgroup are NOT reached (grid < CUs) - and pooled output
    (9, 96, 160, 16, 12, 4, True),        # 3 K slabs, 432 tiles x 3 channel tiles
    (40, 64, 128, 40, 36, 0, False),      # 14400 tiles x 2 channel tiles on 256 persistent workgroups: the cross-tile fetch stream
    (1, 256, 64, 1, 1, 0, False),         # a single 1x1 image: every patch row / column but one is padding
    (2, 64, 64, 2, 7, 0, False),
    (3, 32, 64, 30, 26, 0, True),         # Cin = 32: a tile is a single K slab (the 208x208 layer of Darknet-19), pooled output
    (2, 32, 40, 13, 9, 32, False),        # ... ragged map, channel window, Cout not a multiple of 64
    (33, 128, 200, 30, 26, 0, True),      # 6435 tiles x 4 channel tiles on 512 workgroup slots (gen 3), last channel tile ragged
    (3, 64, 32, 18, 22, 0, False),        # Cout = 32 (the data gradient of the 208x208 layer): gen 3 idles the waves without channels
    (2, 128, 20, 9, 7, 4, False),         # ... Cout < 32, odd map, channel window
])
@pytest.mark.parametrize('gen', [0, 3])
def test_conv_fwd_implicit_is_bit_identical_to_fused(B, cin, cout, H, W, pad_ch, pool, gen):
    """Y2_ALGO_WINOGRAD_IMPLICIT runs the same GEMM and the same output transform as Y2_ALGO_WINOGRAD_FUSED; what changes is where
    B^T d B is computed (registers of the fused kernel's loader instead of wino_input_kernel + memory), with the same operations in
    the same order.  So the two must agree bit for bit: output, pooled output and the fp64 statistics up to the order of the atomics.
    gen = 3: the same pair as wino_fused3_kernel (y2_conv_params.tile = 3; 32-tile x 64-channel units, two workgroups per CU, 16x16x4
    MFMAs): bit-identical to each other, and equal to the second-generation kernels up to the order of the K sum."""
    import _hip
    L, d = _hip.lib(), dev()
    g = torch.Generator().manual_seed(B * 131 + cin + cout + H * W)
    ldx = cin + pad_ch
    xw = torch.randn(B, H, W, ldx, generator=g).to(d)                  # NHWC; the layer reads channels [pad_ch, pad_ch + cin)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(d)
    sc, sh = torch.randn(cout, generator=g).to(d), (torch.randn(cout, generator=g) * 0.1).to(d)
    wp = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(w), _hip.ptr(wp), cout, cin, 3, 0, _hip.stream()), 'pack')
    u = _hip.wino_weight(wp, cout, cin)
    res = {}
    for algo in (2, 3):
        p = _hip.ConvParams()
        p.x, p.w, p.scale, p.shift = xw.data_ptr() + 4 * pad_ch, u.data_ptr(), sc.data_ptr(), sh.data_ptr()
        y = torch.full((B, H, W, cout + 4), -7.0, device=d)
        st = torch.zeros(32 * 2 * cout, dtype=torch.float64, device=d)
        p.y, p.ldy, p.coff, p.stats = y.data_ptr(), cout + 4, 4, st.data_ptr()
        yp = None
        if pool:
            yp = torch.full((B, H // 2, W // 2, cout), -7.0, device=d)
            p.y_pool, p.ldp, p.poff = yp.data_ptr(), cout, 0
        p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.slope, p.algo = B, H, W, cin, ldx, cout, 3, 0.1, algo
        p.tile = gen
        need = _hip.conv_workspace(p, d)
        assert need >= 0
        if algo == 3:
            assert need < 16 * ((H + 1) // 2) * ((W + 1) // 2) * B * cin * 4 or B * H * W < 64      # no transformed input in the workspace
        _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'conv algo %d' % algo)
        torch.cuda.synchronize()
        res[algo] = (y.cpu(), yp.cpu() if yp is not None else None, st.cpu().view(32, -1).sum(0))
    assert torch.equal(res[2][0], res[3][0])
    assert torch.all(res[3][0][..., :4] == -7.0)
    if pool:
        assert torch.equal(res[2][1], res[3][1])
    np.testing.assert_allclose(res[3][2].numpy(), res[2][2].numpy(), rtol=1e-12, atol=1e-9)
    # and it is the right answer (fp64 reference)
    x_nchw = xw[..., pad_ch:].permute(0, 3, 1, 2).cpu()
    z, ref = ref_conv(x_nchw, w.cpu(), sc.cpu(), sh.cpu(), 0.1, 3)
    assert rel_err(res[3][0][..., 4:].permute(0, 3, 1, 2), ref) <= 4 * CONV_TOL
    if pool:
        assert rel_err(res[3][1].permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= 4 * CONV_TOL
    s1, s2 = z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))
    np.testing.assert_allclose(res[3][2][:cout].numpy(), s1.numpy(), rtol=1e-5, atol=2e-5 * float(s2.max().sqrt()))
    np.testing.assert_allclose(res[3][2][cout:].numpy(), s2.numpy(), rtol=2e-5)


@pytest.mark.parametrize('algo', [1, 2, 23])      # 23: the fused algorithm with tile = 3 (third-generation kernel)
@pytest.mark.parametrize('pool', [False, True])
def test_conv_fwd_winograd_batch_chunks(pool, monkeypatch, algo):
    """Y2_WINO_CHUNK_MB bounds the Winograd workspace by running the three stages per batch chunk: 5 images in chunks of 2+2+1
    (ragged last chunk) must give the same output and the same BN statistics as one chunk."""
    tile = 3 if algo > 3 else 0
    algo = algo // 10 if algo > 3 else algo
    B, cin, cout, H, W = (5 if algo == 1 else 13), 64, 96, 14 if pool else 13, 10 if pool else 11      # fused: no product tensor, smaller chunks
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    scale, shift = torch.randn(cout, generator=g), torch.randn(cout, generator=g) * 0.1
    z, ref = ref_conv(x, w, scale, shift, 0.1, 3)
    per_image = 16 * ((H + 1) // 2) * ((W + 1) // 2) * (cin + (cout if algo == 1 else 0)) * 4
    assert per_image < (1 << 20) < B * per_image              # 1 MB holds fewer than the B images of this layer
    monkeypatch.setenv('Y2_WINO_CHUNK_MB', '1')
    out = run_conv(x, w, scale, shift, 0.1, 3, pool=pool, both=pool, wino=algo, stats=True, tile=tile)
    assert rel_err(out['y'].permute(0, 3, 1, 2), ref) <= 4 * CONV_TOL
    if pool:
        assert rel_err(out['y_pool'].permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= 4 * CONV_TOL
    np.testing.assert_allclose(out['stats'][:cout].numpy(), z.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=2e-5 * float((z * z).sum((0, 2, 3)).max().sqrt()))
    np.testing.assert_allclose(out['stats'][cout:].numpy(), (z * z).sum((0, 2, 3)).numpy(), rtol=2e-5)


def _random_conv_cases(n, seed=2024):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < n:
        k = int(rng.choice([1, 3, 3]))
        cin = int(rng.choice([4, 8, 12, 20, 32, 36, 64, 96, 100, 128, 160, 256]))
        cout = int(rng.choice([4, 12, 20, 32, 48, 64, 100, 125, 128, 192, 320]))
        H, W = int(rng.randint(1, 30)), int(rng.randint(1, 30))
        B = int(rng.randint(1, 5))
        pool = bool(rng.rand() < 0.3)
        if pool:
            H, W = 2 * max(1, H // 2), 2 * max(1, W // 2)
        algos = [0]
        if k == 3 and cin % 4 == 0 and cout % 4 == 0:
            algos.append(1)
            if cin % 32 == 0:
                algos.append(2)
        algo = int(rng.choice(algos))
        tile = int(rng.choice([0, 1, 2, 3, 5])) if algo != 2 else 0
        cases.append((B, cin, cout, H, W, k, tile, algo, pool))
    return cases


@pytest.mark.parametrize('B,cin,cout,H,W,k,tile,algo,pool', _random_conv_cases(48))
def test_conv_fwd_random_shapes(B, cin, cout, H, W, k, tile, algo, pool):
    """Seeded random problem shapes (1x1 ... 29x29 maps, channel counts that are not multiples of the tile sizes, every tile
    configuration, direct / Winograd / fused Winograd, with and without the fused pool) against the fp64 reference: output,
    pooled output and the BN statistics."""
    g = torch.Generator().manual_seed(B * 7919 + cin * 31 + cout * 17 + H * 5 + W + k)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale = torch.randn(cout, generator=g)
    shift = torch.randn(cout, generator=g) * 0.1
    z, ref = ref_conv(x, w, scale, shift, 0.1, k)
    out = run_conv(x, w, scale, shift, 0.1, k, tile=tile, pool=pool, both=pool, wino=algo, stats=True)
    tol = CONV_TOL * (4 if algo else 1)
    assert rel_err(out['y'].permute(0, 3, 1, 2), ref) <= tol
    if pool:
        assert rel_err(out['y_pool'].permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= tol
    s1, s2 = z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))
    np.testing.assert_allclose(out['stats'][:cout].numpy(), s1.numpy(), rtol=1e-5, atol=2e-5 * float(s2.max().sqrt()))
    np.testing.assert_allclose(out['stats'][cout:].numpy(), s2.numpy(), rtol=2e-5)


@pytest.mark.parametrize('tile', [1, 2, 3, 5, 7, 8, 9])
@pytest.mark.parametrize('both', [False, True])
def test_conv_fwd_fused_maxpool(tile, both):
    g = torch.Generator().manual_seed(5)
    B, cin, cout, H, W, k = 2, 32, 48, 12, 20, 3
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    scale = torch.randn(cout, generator=g)   # negative scales: pool must come AFTER the affine
    shift = torch.randn(cout, generator=g) * 0.1
    _, ref = ref_conv(x, w, scale, shift, 0.1, k)
    out = run_conv(x, w, scale, shift, 0.1, k, pool=True, both=both, tile=tile)
    assert rel_err(out['y_pool'].permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= CONV_TOL
    if both:
        assert rel_err(out['y'].permute(0, 3, 1, 2), ref) <= CONV_TOL


def test_conv_fwd_reorg_concat_write_through():
    g = torch.Generator().manual_seed(6)
    B, cin, cout, H, W = 2, 64, 8, 6, 10
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * 0.2
    _, ref = ref_conv(x, w, None, None, 0.1, 1)
    out = run_conv(x, w, None, None, 0.1, 1, reorg=True, coff=0, extra=5)
    y = out['y']
    assert torch.all(y[..., 4 * cout:] == -7.0)
    assert rel_err(y[..., :4 * cout].permute(0, 3, 1, 2), odark.reorg(ref)) <= CONV_TOL


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 3, 32, 32, 64), (1, 3, 32, 20, 36), (2, 3, 40, 16, 32), (1, 4, 7, 18, 34)])
def test_conv0_matches_fp64_reference(B, cin, cout, H, W):
    import _hip
    L = _hip.lib()
    d = dev()
    g = torch.Generator().manual_seed(cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.3
    scale = torch.randn(cout, generator=g)
    shift = torch.randn(cout, generator=g) * 0.1
    z, ref = ref_conv(x, w, scale, shift, 0.1, 3)
    y = torch.empty(B, H, W, cout, device=d)
    yp = torch.empty(B, H // 2, W // 2, cout, device=d)
    st = torch.zeros(32 * 2 * cout, dtype=torch.float64, device=d)
    xd, wd, sc, sh = x.to(d), w.to(d), scale.to(d), shift.to(d)
    _hip.check(L.y2_conv0_fwd(_hip.ptr(xd), _hip.ptr(wd), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(y), _hip.ptr(yp), _hip.ptr(st),
                              B, H, W, cin, cout, cout, cout, 0.1, _hip.stream()), 'conv0')
    torch.cuda.synchronize()
    st = st.view(32, -1).sum(0)
    assert rel_err(y.permute(0, 3, 1, 2), ref) <= CONV_TOL
    assert rel_err(yp.permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= CONV_TOL
    np.testing.assert_allclose(st[:cout].cpu().numpy(), z.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st[cout:].cpu().numpy(), (z * z).sum((0, 2, 3)).numpy(), rtol=1e-5)


@pytest.mark.parametrize('B,H,W,cout', [(9, 416, 416, 32), (40, 208, 224, 32), (3, 330, 500, 16)])
def test_conv0_persistent_workgroups_walk_several_tiles(B, H, W, cout):
    """More 16 x 32-pixel tiles than the persistent grid has workgroups (2,048): a workgroup computes its 2nd, 3rd ... tile out of the patches it staged two tiles
    ahead (three LDS buffers, csrc/conv0_fwd.hip).  Inference form (pooled output only), training form (raw z + statistics) and - ragged size, 16 channels -
    the general kernel against the fp64 reference."""
    import _hip
    L = _hip.lib()
    d = dev()
    g = torch.Generator().manual_seed(B + W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    scale, shift = torch.randn(cout, generator=g), torch.randn(cout, generator=g) * 0.1
    z, ref = ref_conv(x, w, scale, shift, 0.1, 3)
    xd, wd, sc, sh = x.to(d), w.to(d), scale.to(d), shift.to(d)
    yp = torch.empty(B, H // 2, W // 2, cout, device=d)
    _hip.check(L.y2_conv0_fwd(_hip.ptr(xd), _hip.ptr(wd), _hip.ptr(sc), _hip.ptr(sh), None, _hip.ptr(yp), None, B, H, W, 3, cout, 0, cout, 0.1, _hip.stream()), 'pool')
    assert rel_err(yp.permute(0, 3, 1, 2), F.max_pool2d(ref, 2)) <= CONV_TOL
    zz = torch.empty(B, H, W, cout, device=d)
    st = torch.zeros(32 * 2 * cout, dtype=torch.float64, device=d)
    _hip.check(L.y2_conv0_fwd(_hip.ptr(xd), _hip.ptr(wd), None, None, _hip.ptr(zz), None, _hip.ptr(st), B, H, W, 3, cout, cout, 0, 1.0, _hip.stream()), 'train')
    torch.cuda.synchronize()
    assert rel_err(zz.permute(0, 3, 1, 2), z) <= CONV_TOL
    st = st.view(32, -1).sum(0).cpu()
    np.testing.assert_allclose(st[:cout].numpy(), z.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(st[cout:].numpy(), (z * z).sum((0, 2, 3)).numpy(), rtol=1e-5)


@pytest.mark.parametrize('cout', [32, 64])
def test_conv0_specialised_epilogues_equal_the_general_kernel(cout):
    """The tile-aligned first layer runs straight-line variants of conv0_kernel (csrc/conv0_fwd.hip launch0): pooled-only takes the 2x2 max BEFORE affine + LeakyReLU
    with the sign of the scale folded into the weights (claimed exact, scales of both signs here), the training forward stores raw z with packed statistics.
    All outputs of one call (out mask 7) come from the general kernel: the reference for every variant."""
    import _hip
    L = _hip.lib()
    d = dev()
    g = torch.Generator().manual_seed(17 + cout)
    B, H, W = 3, 48, 96
    x = torch.randn(B, 3, H, W, generator=g).to(d)
    w = (torch.randn(cout, 3, 3, 3, generator=g) * 0.3).to(d)
    sc = torch.randn(cout, generator=g).to(d)
    sh = (torch.randn(cout, generator=g) * 0.1).to(d)
    assert (sc < 0).any() and (sc > 0).any()

    def call(scale, shift, want_y, want_pool, want_stats, slope):
        y = torch.full((B, H, W, cout), -7.0, device=d) if want_y else None
        yp = torch.full((B, H // 2, W // 2, cout), -7.0, device=d) if want_pool else None
        st = torch.zeros(32 * 2 * cout, dtype=torch.float64, device=d) if want_stats else None
        _hip.check(L.y2_conv0_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(scale), _hip.ptr(shift), _hip.ptr(y), _hip.ptr(yp), _hip.ptr(st),
                                  B, H, W, 3, cout, cout if want_y else 0, cout if want_pool else 0, slope, _hip.stream()), 'conv0')
        torch.cuda.synchronize()
        return y, yp, (st.view(32, -1).sum(0) if want_stats else None)

    for slope in (0.1, 1.0, 0.0):
        y7, p7, _ = call(sc, sh, True, True, True, slope)
        _, p2, _ = call(sc, sh, False, True, False, slope)             # pool-first
        assert torch.equal(p2, p7), 'pooled-only (slope %g)' % slope
        y1, _, _ = call(sc, sh, True, False, False, slope)
        assert torch.equal(y1, y7)
    _, pneg, _ = call(sc, sh, False, True, False, -0.5)                # a non-monotone activation: the general kernel
    y7n, p7n, _ = call(sc, sh, True, True, True, -0.5)
    assert torch.equal(pneg, p7n)
    # training forward: raw z + statistics
    z7, _, st7 = call(None, None, True, True, True, 1.0)
    z5, _, st5 = call(None, None, True, False, True, 1.0)
    assert torch.equal(z5, z7)
    np.testing.assert_allclose(st5.cpu().numpy(), st7.cpu().numpy(), rtol=2e-6, atol=1e-4)
    ref = z7.double()
    np.testing.assert_allclose(st5[:cout].cpu().numpy(), ref.sum((0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st5[cout:].cpu().numpy(), (ref * ref).sum((0, 1, 2)).cpu().numpy(), rtol=1e-5)


def test_maxpool2():
    import _hip
    for C in (16, 6):   # 16-B vector path and scalar path (pruned widths)
        x = torch.randn(2, C, 8, 12)
        xd = to_nhwc(x).to(dev())
        y = torch.empty(2, 4, 6, C, device=dev())
        _hip.check(_hip.lib().y2_maxpool2_fwd(_hip.ptr(xd), _hip.ptr(y), 2, 8, 12, C, C, C, _hip.stream()), 'pool')
        assert torch.equal(y.cpu().permute(0, 3, 1, 2), F.max_pool2d(x, 2))


# ------------------------------------------------------------------ the plugin end to end
def make_plugin(sd, num_cls=20, bn=True):
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1' if bn else '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, num_cls)
    dnn.load_state_dict(sd, strict=False)
    inf = model.Inference(cfg, dnn, anchors)
    return inf.to(dev()).eval()


def test_darknet_narrow_matches_reference_fixture(golden):
    import model
    g = golden('forward_narrow')
    sd = odark.init_state_dict(5, 20, seed=0, channels=NARROW, head_scale=1 / 8.0)
    inf = make_plugin(sd)
    x = synth.images(2, 96, seed=1)
    with torch.no_grad():
        pred = model._inference(inf, x.to(dev()))
        f64 = odark.forward(x.double(), {k: v.double() for k, v in sd.items()})
    assert tuple(pred['feature'].shape) == g['feature'].shape
    assert rel_err(pred['feature'], f64) <= CONV_TOL
    assert rel_err(pred['feature'], torch.from_numpy(g['feature'])) <= 2 * CONV_TOL   # the reference's own fp32 output
    for k in ('iou', 'center_offset', 'size_norm', 'yx_min', 'yx_max', 'logits'):
        np.testing.assert_allclose(pred[k].cpu().numpy(), g[k], rtol=2e-4, atol=2e-4)


def test_darknet_full_width_matches_reference_fixture(golden):
    g = golden('forward_full')
    sd = odark.init_state_dict(5, 20, seed=0, head_scale=1 / 40.0)
    inf = make_plugin(sd)
    with torch.no_grad():
        f = inf.dnn(synth.images(1, 64, seed=1).to(dev()))
    assert rel_err(f, torch.from_numpy(g['feature_fp64'])) <= CONV_TOL


def test_darknet_no_batchnorm_variant():
    sd = odark.init_state_dict(5, 20, seed=3, channels=NARROW, bn=False, head_scale=1 / 8.0)
    inf = make_plugin(sd, bn=False)
    x = synth.images(1, 64, seed=4)
    with torch.no_grad():
        f = inf.dnn(x.to(dev()))
        f64 = odark.forward(x.double(), {k: v.double() for k, v in sd.items()})
    assert rel_err(f, f64) <= CONV_TOL


@pytest.mark.parametrize('S,B', [(416, 2), (320, 1), (608, 1)])
def test_darknet19_full_size_every_layer(S, B):
    """Config-2 shape (and the multi-scale extremes): every conv block's output against the fp32 oracle taps, final
    feature against fp64."""
    sd = odark.init_state_dict(5, 20, seed=0, head_scale=1 / 40.0)
    inf = make_plugin(sd)
    x = synth.images(B, S, seed=1)
    with torch.no_grad():
        f = inf.dnn(x.to(dev()))
        torch.set_num_threads(max(1, torch.get_num_threads()))
        f32 = odark.forward(x, sd)
    assert tuple(f.shape) == (B, 125, S // 32, S // 32)
    assert rel_err(f, f32) <= 4 * CONV_TOL   # fp32 oracle itself carries ~5e-6 rms of summation-order noise


# ------------------------------------------------------------------ decode / filter / IoU / NMS
@pytest.mark.parametrize('name', ['decode_voc', 'decode_coco', 'decode_1cls'])
def test_decode_matches_reference_fixture(golden, name):
    import model
    g = golden(name)
    feat = torch.from_numpy(g['feature']).to(dev())
    d = model.decode(to_nhwc(feat), torch.from_numpy(synth.ANCHORS_VOC), 5, want_prob=True)
    for k in ('iou', 'center_offset', 'yx_min', 'yx_max'):
        np.testing.assert_allclose(d[k].cpu().numpy(), g[k], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(d['size_norm'].cpu().numpy(), g['size_norm'])
    if 'logits' in g.files:
        prob = torch.softmax(torch.from_numpy(g['logits']), -1)
        np.testing.assert_allclose(d['prob'].cpu().numpy(), prob.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_array_equal(d['cls'].cpu().numpy(), prob.argmax(-1).numpy())
        np.testing.assert_allclose(d['prob_cls'].cpu().numpy(), prob.max(-1)[0].numpy(), rtol=1e-5)
    else:
        assert torch.all(d['prob_cls'] == 1) and torch.all(d['cls'] == 0)


def boxes_np(b):
    b = np.array(b, np.float32)
    return b[:, :2].copy(), b[:, 2:].copy()


def test_iou_known_answers_and_bit_exact_random():
    import utils.iou.torch as iou
    from test_oracle import KNOWN
    d = dev()
    for b1, b2, ans in KNOWN:   # the reference's own known-answer cases (utils/iou/torch.py:79-113)
        m = iou.iou_matrix(*(torch.from_numpy(t).to(d) for t in boxes_np(b1) + boxes_np(b2)))
        np.testing.assert_almost_equal(m.cpu().numpy(), np.array(ans, np.float32))
    rng = np.random.RandomState(0)
    c1, s1 = rng.uniform(0, 13, (3, 211, 2)).astype(np.float32), rng.uniform(0, 6, (3, 211, 2)).astype(np.float32)
    c2, s2 = rng.uniform(0, 13, (3, 37, 2)).astype(np.float32), rng.uniform(0, 6, (3, 37, 2)).astype(np.float32)
    s2[:, -3:] = 0   # degenerate (padded GT rows)
    args = (c1 - s1 / 2, c1 + s1 / 2, c2 - s2 / 2, c2 + s2 / 2)
    targs = tuple(torch.from_numpy(a).to(d) for a in args)
    np.testing.assert_array_equal(iou.batch_iou_matrix(*targs).cpu().numpy(), oiou.batch_iou_matrix(*args))
    np.testing.assert_array_equal(iou.iou_matrix(*(t[0] for t in targs)).cpu().numpy(), oiou.iou_matrix(*(a[0] for a in args)))
    np.testing.assert_array_equal(iou.batch_intersection_area(*targs).cpu().numpy(), oiou.batch_intersection_area(*args))
    pair = (args[0][:, :37], args[1][:, :37], args[2], args[3])
    np.testing.assert_array_equal(iou.batch_iou_pair(*(torch.from_numpy(a).to(d) for a in pair)).cpu().numpy(), oiou.batch_iou_pair(*pair))


@pytest.mark.parametrize('n', [0, 1, 2, 50, 200, 845, 2000])
def test_nms_bit_exact_vs_reference_fixture(golden, n):
    import utils.postprocess as post
    g = golden('nms')
    score, mn, mx = synth.nms_boxes(n)
    d = dev()
    for ov in (0.45, 0.5):
        keep = post.nms(torch.from_numpy(score).to(d), torch.from_numpy(mn).to(d).view(-1, 2), torch.from_numpy(mx).to(d).view(-1, 2), ov)
        assert keep == g['n%d_ov%d' % (n, int(ov * 100))].tolist()


def test_nms_ties_limits_and_batch():
    import utils.postprocess as post
    d = dev()
    rng = np.random.RandomState(9)
    B, stride = 5, 700
    ns = [700, 0, 1, 333, 64]
    score = rng.randint(0, 40, (B, stride)).astype(np.float32) / 40     # many exact ties
    c, s = rng.uniform(0, 13, (B, stride, 2)).astype(np.float32), rng.uniform(0.5, 5, (B, stride, 2)).astype(np.float32)
    mn, mx = c - s / 2, c + s / 2
    for limit in (200, 64, 1000):
        keep, cnt = post.nms_batch(torch.from_numpy(score).to(d), torch.from_numpy(mn).to(d), torch.from_numpy(mx).to(d),
                                   torch.tensor(ns, dtype=torch.int32, device=d), 0.45, limit)
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        for b in range(B):
            ref = onms.nms(score[b, :ns[b]], mn[b, :ns[b]], mx[b, :ns[b]], 0.45, limit)
            assert keep[b, :cnt[b]].tolist() == ref


def test_postprocess_matches_reference_fixture(golden):
    import detect
    dd = golden('decode_voc')
    g = golden('postprocess')
    d = dev()
    for fix in (0, 1):
        cfg = configparser.ConfigParser()
        cfg.read_dict({'detect': {'threshold': '0.3', 'threshold_cls': '0.005', 'fix': str(fix), 'overlap': '0.45'}})
        for b in range(dd['iou'].shape[0]):
            prob = torch.softmax(torch.from_numpy(dd['logits'][b]), -1).view(-1, 20)
            r = detect.postprocess(cfg, torch.from_numpy(dd['iou'][b]).view(-1).to(d), torch.from_numpy(dd['yx_min'][b]).view(-1, 2).to(d),
                                   torch.from_numpy(dd['yx_max'][b]).view(-1, 2).to(d), prob.to(d))
            tag = 'fix%d_b%d_' % (fix, b)
            assert bool(g[tag + 'none']) == (r is None)
            if r is not None:
                for name, t in zip(('iou', 'yx_min', 'yx_max', 'cls', 'score'), r):
                    np.testing.assert_array_equal(t.cpu().numpy(), g[tag + name])   # identical inputs -> bit-exact survivors


def test_detect_batch_end_to_end_vs_oracle(golden):
    """Device-resident decode -> filter -> NMS on the head image; survivors must equal the oracle's run on the
    GPU-decoded values (bit-exact given identical inputs)."""
    import detect
    g = golden('decode_voc')
    feat = to_nhwc(torch.from_numpy(g['feature'])).to(dev())
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    for fix in (False, True):
        d = detect.detect_batch(feat, anchors, fix=fix)
        res = detect.postprocess_batch(d, fix=fix)
        B = feat.size(0)
        iou = d['iou'].view(B, -1).cpu().numpy()
        mn, mx = d['yx_min'].view(B, -1, 2).cpu().numpy(), d['yx_max'].view(B, -1, 2).cpu().numpy()
        prob = d['prob'].view(B, iou.shape[1], -1).cpu().numpy()
        for b in range(B):
            ref = odet.postprocess(iou[b], mn[b], mx[b], prob[b], fix=fix)
            assert (ref is None) == (res[b] is None)
            if ref is not None:
                for got, want in zip(res[b], ref[:5]):
                    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('threshold', [1.5, 0.0])
def test_detect_batch_nothing_visible_and_everything_visible(golden, threshold):
    """threshold above every score -> every image yields None (detect.py:69 returns nothing to draw); threshold 0 -> all
    845 cells x anchors are candidates and NMS runs on the full list (limit 200): still equal to the oracle."""
    import detect
    g = golden('decode_voc')
    feat = to_nhwc(torch.from_numpy(g['feature'])).to(dev())[:2].contiguous()
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    d = detect.detect_batch(feat, anchors, fix=False, threshold=threshold)
    res = detect.postprocess_batch(d, fix=False)
    B = feat.size(0)
    iou = d['iou'].view(B, -1).cpu().numpy()
    mn, mx = d['yx_min'].view(B, -1, 2).cpu().numpy(), d['yx_max'].view(B, -1, 2).cpu().numpy()
    prob = d['prob'].view(B, iou.shape[1], -1).cpu().numpy()
    for b in range(B):
        ref = odet.postprocess(iou[b], mn[b], mx[b], prob[b], fix=False, threshold=threshold)
        if threshold > 1:
            assert ref is None and res[b] is None and int(d['count'][b]) == 0
        else:
            assert int(d['count'][b]) == iou.shape[1]
            for got, want in zip(res[b], ref[:5]):
                np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('rows,cols,C,B,fix,thr,overlap,limit,seed', [
    (10, 10, 20, 3, False, 0.3, 0.45, 200, 1), (13, 13, 80, 2, True, 0.005, 0.45, 200, 2), (19, 19, 20, 2, False, 0.5, 0.3, 50, 3),
    (13, 13, 0, 4, False, 0.2, 0.6, 200, 4), (19, 19, 80, 1, True, 0.02, 0.45, 1000, 5), (7, 12, 3, 2, False, 0.1, 0.5, 10, 6)])
def test_detect_batch_random_configurations_vs_oracle(rows, cols, C, B, fix, thr, overlap, limit, seed):
    """decode -> filter -> NMS -> gather over grid sizes (incl. a non-square one), class counts (0 = single class), both filter
    modes, thresholds, overlaps and keep limits: survivors bit-exact against the oracle run on the GPU-decoded values."""
    import detect
    A = 5
    g = torch.Generator().manual_seed(500 + seed)
    ch = A * (5 + C)
    feat = (torch.randn(B, rows, cols, ch, generator=g) * 1.5).to(dev())
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    kw = dict(threshold_cls=thr) if fix else dict(threshold=thr)
    d = detect.detect_batch(feat, anchors, fix=fix, overlap=overlap, limit=limit, **kw)
    res = detect.postprocess_batch(d, fix=fix, **(dict(threshold_cls=thr) if fix else {}))
    n = rows * cols * A
    iou = d['iou'].view(B, n).cpu().numpy()
    mn, mx = d['yx_min'].view(B, n, 2).cpu().numpy(), d['yx_max'].view(B, n, 2).cpu().numpy()
    prob = d['prob'].view(B, n, -1).cpu().numpy()
    for b in range(B):
        ref = odet.postprocess(iou[b], mn[b], mx[b], prob[b], fix=fix, overlap=overlap, limit=limit,
                               **(dict(threshold_cls=thr) if fix else dict(threshold=thr)))
        assert (ref is None) == (res[b] is None)
        if ref is not None:
            for got, want in zip(res[b], ref[:5]):
                np.testing.assert_array_equal(got.cpu().numpy(), want)
    # the same detections handed over on the host (one copy per result buffer instead of five per image): identical values
    host = detect.postprocess_batch(d, fix=fix, to_host=True, **(dict(threshold_cls=thr) if fix else {}))
    for b in range(B):
        assert (host[b] is None) == (res[b] is None)
        if host[b] is not None:
            for got, want in zip(host[b], res[b]):
                assert not got.is_cuda and torch.equal(got, want.cpu())


# ------------------------------------------------------------------ general convolution (ResNet plugin: model/resnet.py)
GEN_CASES = [
    # B, Cin, Cout, H, W, k, stride, pad, residual, tile
    (2, 4, 64, 32, 40, 7, 2, 3, False, 0),      # stem 7x7 s2 p3 on the zero-padded 4-channel NHWC input
    (2, 64, 64, 19, 19, 3, 2, 1, False, 0),     # 3x3 s2 (Bottleneck.conv2 of a down-sampling block)
    (2, 64, 128, 20, 20, 1, 2, 0, False, 3),    # 1x1 s2 down-sample branch
    (2, 64, 256, 10, 10, 1, 1, 0, True, 0),     # 1x1 + residual + ReLU
    (1, 16, 24, 9, 11, 3, 1, 1, True, 5),       # small Cin through the linear-K path
    (2, 8, 40, 12, 12, 5, 1, 0, False, 2),      # 5x5 valid padding
]


@pytest.mark.parametrize('B,cin,cout,H,W,k,stride,pad,residual,tile', GEN_CASES)
def test_conv_general_stride_kernel_residual(B, cin, cout, H, W, k, stride, pad, residual, tile):
    import _hip
    L = _hip.lib()
    d = dev()
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    z = F.conv2d(x.double(), w.double(), stride=stride, padding=pad) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    Ho, Wo = z.shape[-2:]
    res = torch.randn(B, cout, Ho, Wo, generator=g) if residual else None
    if residual:
        z = z + res.double()
    ref = torch.relu(z)
    xd = to_nhwc(x).to(d)
    wd = w.to(d).contiguous()
    wp = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(wd), _hip.ptr(wp), cout, cin, k, 0, _hip.stream()), 'pack')
    y = torch.full((B, Ho, Wo, cout + 3), -7.0, device=d)
    sc, sh = scale.to(d), shift.to(d)
    rd = to_nhwc(res).to(d) if residual else None
    p = _hip.ConvParams()
    p.x, p.w, p.scale, p.shift, p.y = xd.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr()
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize = B, H, W, cin, cin, cout, k
    p.ldy, p.slope, p.tile = cout + 3, 0.0, tile
    p.stride, p.pad_plus1 = stride, pad + 1
    if residual:
        p.residual, p.ldr = rd.data_ptr(), cout
    _hip.conv_workspace(p, d)
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'conv')
    torch.cuda.synchronize()
    assert torch.all(y[..., cout:] == -7.0)
    assert rel_err(y[..., :cout].permute(0, 3, 1, 2), ref) <= CONV_TOL


# ------------------------------------------------------------------ ResNet plugin (BASELINE config 5: model-plugin swap)
def make_resnet(arch, sd, num_cls):
    import model
    import model.resnet
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}, 'model': {'pretrained': '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    net = getattr(model.resnet, arch)(model.ConfigChannels(cfg, sd), anchors, num_cls)
    net.load_state_dict(sd, strict=False)
    return model.Inference(cfg, net, anchors).to(dev()).eval()


@pytest.mark.parametrize('arch,width,S,C,B', [('resnet50', 8, 96, 80, 2), ('resnet18', 8, 64, 20, 2), ('resnet50', 64, 64, 80, 1)])
def test_resnet_matches_reference_fixture(golden, arch, width, S, C, B):
    from oracle import resnet as ores
    g = golden('resnet')
    sd = ores.init_state_dict(arch, 5, C, seed=0, width=width, head_scale=0.25)
    inf = make_resnet(arch, sd, C)
    with torch.no_grad():
        f = inf.dnn(synth.images(B, S, seed=1).to(dev()))
    assert tuple(f.shape) == g['%s_w%d_feature' % (arch, width)].shape
    assert rel_err(f, torch.from_numpy(g['%s_w%d_fp64' % (arch, width)])) <= CONV_TOL
    assert rel_err(f, torch.from_numpy(g['%s_w%d_feature' % (arch, width)])) <= 2 * CONV_TOL


def test_resnet50_608_coco_full_pipeline():
    """Config-5 shape: ResNet-50, 608x608, COCO-80 head -> decode -> filter -> NMS, against the fp32 oracle."""
    import detect
    import model
    from oracle import resnet as ores
    sd = ores.init_state_dict('resnet50', 5, 80, seed=0, head_scale=0.25)
    inf = make_resnet('resnet50', sd, 80)
    x = synth.images(1, 608, seed=1)
    with torch.no_grad():
        pred = model._inference(inf, x.to(dev()))
        f32 = ores.forward(x, sd, 'resnet50')
    assert tuple(pred['feature'].shape) == (1, 425, 19, 19)
    assert rel_err(pred['feature'], f32) <= 4 * CONV_TOL
    d = detect.detect_batch(pred['feature'].permute(0, 2, 3, 1).contiguous(), torch.from_numpy(synth.ANCHORS_VOC), fix=True)
    assert int(d['keep_count'][0]) > 0


# ------------------------------------------------------------------ tiny-yolo plugin and the eval matcher
@pytest.mark.parametrize('div,S,B', [(8, 96, 2), (1, 64, 1)])
def test_tiny_matches_reference_fixture(golden, div, S, B):
    import model
    import model.yolo2
    g = golden('tiny')
    sd = odark.init_tiny_state_dict(5, 20, seed=0, div=div, head_scale=0.25)
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    net = model.yolo2.Tiny(model.ConfigChannels(cfg, sd), anchors, 20)
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    net = net.to(dev()).eval()
    with torch.no_grad():
        f = net(synth.images(B, S, seed=1).to(dev()))
    assert rel_err(f, torch.from_numpy(g['tiny_div%d_fp64' % div])) <= CONV_TOL


def test_eval_matching_like_reference():
    import importlib
    ev = importlib.import_module('eval')
    rng = np.random.RandomState(3)
    c, s = rng.uniform(2, 11, (40, 2)).astype(np.float32), rng.uniform(1, 4, (40, 2)).astype(np.float32)
    gt_min, gt_max = c[:7] - s[:7] / 2, c[:7] + s[:7] / 2
    jit = rng.uniform(-0.4, 0.4, (40, 2)).astype(np.float32)
    pc = np.concatenate([c[:7] + jit[:7], c[:7] + 2 * jit[7:14], c[14:]])
    p_min, p_max = pc - s[:len(pc)] / 2, pc + s[:len(pc)] / 2
    d = dev()
    tp = ev.matching(torch.from_numpy(gt_min).to(d), torch.from_numpy(gt_max).to(d), torch.from_numpy(p_min).to(d), torch.from_numpy(p_max).to(d), 0.5)
    m = oiou.iou_matrix(p_min, p_max, gt_min, gt_max)     # eval.py:67-75 restated on the oracle IoU
    iou, index = m.max(-1), m.argmax(-1)
    want = ev._matching(iou > 0.5, index)
    np.testing.assert_array_equal(tp, want)
    assert tp.sum() >= 5
    assert ev.matching(torch.zeros(0, 2, device=d), torch.zeros(0, 2, device=d), torch.from_numpy(p_min).to(d), torch.from_numpy(p_max).to(d), 0.5).sum() == 0


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 64, 32, 13, 13), (1, 128, 64, 26, 26), (3, 32, 48, 7, 10), (2, 512, 256, 13, 13), (1, 16, 20, 4, 4), (2, 8, 12, 1, 3),
                                            (11, 16, 24, 13, 13), (9, 8, 8, 13, 11), (16, 12, 8, 26, 26)])      # (the last three: mosaic tile grids - 2 x 6 with a missing image, 3 x 3, 4 x 4)
def test_conv_fwd_winograd_f43_for_gradients(B, cin, cout, H, W):
    """Y2_ALGO_WINOGRAD_F43 (Winograd F(4x4,3x3), three kernels, 36 GEMMs; offered to the training step's data gradients): against the fp64
    reference at the single-Winograd-layer tolerance (its larger transform constants: ~1e-5 x rms instead of ~2e-6), ragged 4x4 tiles,
    channel window and affine + LeakyReLU epilogue."""
    import _hip
    L, d = _hip.lib(), dev()
    g = torch.Generator().manual_seed(B + cin + cout + H * W)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    scale, shift = torch.randn(cout, generator=g), torch.randn(cout, generator=g) * 0.1
    _, ref = ref_conv(x, w, scale, shift, 0.1, 3)
    xd = to_nhwc(x).to(d)
    wp = torch.empty(w.numel(), device=d)
    _hip.check(L.y2_pack_weight(_hip.ptr(w.to(d).contiguous()), _hip.ptr(wp), cout, cin, 3, 0, _hip.stream()), 'pack')
    u6 = _hip.wino6_weight(wp, cout, cin)
    sc, sh = scale.to(d), shift.to(d)
    y = torch.full((B, H, W, cout + 8), -7.0, device=d)
    p = _hip.ConvParams()
    p.x, p.w, p.scale, p.shift, p.y = xd.data_ptr(), u6.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr()
    p.B, p.H, p.W, p.Cin, p.ldx, p.Cout, p.ksize, p.ldy, p.coff, p.slope, p.algo, p.tile = B, H, W, cin, cin, cout, 3, cout + 8, 4, 0.1, 6, 0
    assert _hip.conv_workspace(p, d) >= 0
    _hip.check(L.y2_conv_fwd(ctypes.byref(p), _hip.stream()), 'conv algo 6')
    torch.cuda.synchronize()
    yc = y.cpu()
    assert torch.all(yc[..., :4] == -7.0) and torch.all(yc[..., 4 + cout:] == -7.0), 'wrote outside its channel window'
    err = rel_err(yc[..., 4:4 + cout].permute(0, 3, 1, 2), ref)
    assert err <= 4 * CONV_TOL, err
    p.y_pool = y.data_ptr()                      # no pooled output / statistics in this algorithm: refused, not ignored
    assert L.y2_conv_fwd(ctypes.byref(p), _hip.stream()) != 0
