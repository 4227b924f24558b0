"""Pin the CPU oracle: the reference's own IoU known-answer tests, and the
fixtures produced by running the reference itself (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import darknet as odark
from oracle import detect as odet
from oracle import head as ohead
from oracle import iou as oiou
from oracle import loss as oloss
from oracle import nms as onms
from oracle import synth
from oracle.make_golden import NARROW


# ---- the reference's known-answer IoU tests (utils/iou/torch.py:79-113, utils/iou/numpy.py:108-142)
CENTER = [(1, 1, 2, 2)]
AROUND = [(0, 0, 1, 1), (0, 1, 1, 2), (0, 2, 1, 3), (1, 0, 2, 1), (2, 0, 3, 1), (1, 2, 2, 3), (2, 1, 3, 2), (2, 2, 3, 3)]
BIG = [(1, 1, 3, 3), (0, 0, 4, 4)]
QUAD = [(0, 0, 2, 2), (2, 0, 4, 2), (0, 2, 2, 4), (2, 2, 4, 4)]
KNOWN = [(CENTER, AROUND, [[0] * 8]), (BIG, QUAD, [[1 / 7] * 4, [4 / 16] * 4])]


def split(b):
    b = np.array(b, np.float32)
    return b[:, :2], b[:, 2:]


@pytest.mark.parametrize('b1,b2,ans', KNOWN)
def test_iou_matrix_known(b1, b2, ans):
    m = oiou.iou_matrix(*split(b1), *split(b2))
    np.testing.assert_almost_equal(m, np.array(ans, np.float32))


@pytest.mark.parametrize('b1,b2,ans', KNOWN)
def test_batch_iou_matrix_known(b1, b2, ans):
    # utils/iou/torch.py:179-213: rows permuted per batch item
    rng = np.random.RandomState(0)
    mn1, mx1 = split(b1)
    mn2, mx2 = split(b2)
    ans = np.array(ans, np.float32)
    p1 = [rng.permutation(len(b1)) for _ in range(3)]
    p2 = [rng.permutation(len(b2)) for _ in range(3)]
    m = oiou.batch_iou_matrix(np.stack([mn1[p] for p in p1]), np.stack([mx1[p] for p in p1]),
                              np.stack([mn2[p] for p in p2]), np.stack([mx2[p] for p in p2]))
    for b in range(3):
        np.testing.assert_almost_equal(m[b], ans[p1[b]][:, p2[b]])


def test_batch_iou_pair_known():
    # utils/iou/torch.py:255-289: pairs (a_i, b_i)
    mn1, mx1 = split(BIG * 2)
    mn2, mx2 = split(QUAD)
    m = oiou.batch_iou_pair(mn1[None], mx1[None], mn2[None], mx2[None])
    np.testing.assert_almost_equal(m[0], np.array([1 / 7, 4 / 16, 1 / 7, 4 / 16], np.float32))


# ---- fixtures generated from the reference
def test_nms_matches_reference(golden):
    g = golden('nms')
    for n in (0, 1, 2, 50, 200, 845, 2000):
        score, mn, mx = synth.nms_boxes(n)
        for ov in (0.45, 0.5):
            keep = onms.nms(score, mn, mx, ov)
            assert keep == g['n%d_ov%d' % (n, int(ov * 100))].tolist()


@pytest.mark.parametrize('name,A', [('decode_voc', 5), ('decode_coco', 5), ('decode_1cls', 5)])
def test_decode_matches_reference(golden, name, A):
    g = golden(name)
    pred = ohead.decode(torch.from_numpy(g['feature']), torch.from_numpy(synth.ANCHORS_VOC[:A]))
    for k in ('iou', 'center_offset', 'size_norm', 'yx_min', 'yx_max'):
        np.testing.assert_array_equal(pred[k].numpy(), g[k])  # same torch kernels -> exact
    if 'logits' in g.files:
        np.testing.assert_array_equal(pred['logits'].numpy(), g['logits'])
    else:
        assert 'logits' not in pred


def test_forward_narrow_matches_reference(golden):
    g = golden('forward_narrow')
    sd = odark.init_state_dict(5, 20, seed=0, channels=NARROW, head_scale=1 / 8.0)
    with torch.no_grad():
        f = odark.forward(synth.images(2, 96, seed=1), sd)
    rms = float(np.sqrt((g['feature'] ** 2).mean()))
    assert np.abs(f.numpy() - g['feature']).max() <= 2e-5 * rms


def test_forward_full_matches_reference(golden):
    g = golden('forward_full')
    sd = odark.init_state_dict(5, 20, seed=0, head_scale=1 / 40.0)
    x = synth.images(1, 64, seed=1)
    with torch.no_grad():
        f = odark.forward(x, sd)
        f64 = odark.forward(x.double(), {k: v.double() for k, v in sd.items()})
    rms = float(np.sqrt((g['feature_fp64'] ** 2).mean()))
    assert np.abs(f64.numpy() - g['feature_fp64']).max() <= 1e-9 * rms
    assert np.abs(f.numpy() - g['feature_fp64']).max() <= 2e-5 * rms


def test_reorg_layout():
    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).view(2, 3, 4, 6)
    y = odark.reorg(x)
    for dy in range(2):
        for dx in range(2):
            for c in range(3):
                assert torch.equal(y[:, (dy * 2 + dx) * 3 + c], x[:, c, dy::2, dx::2])


def test_postprocess_matches_reference(golden):
    d = golden('decode_voc')
    g = golden('postprocess')
    for fix in (0, 1):
        for b in range(d['iou'].shape[0]):
            prob = torch.softmax(torch.from_numpy(d['logits'][b]), -1).numpy().reshape(-1, 20)
            r = odet.postprocess(d['iou'][b].reshape(-1), d['yx_min'][b].reshape(-1, 2), d['yx_max'][b].reshape(-1, 2), prob, fix=bool(fix))
            tag = 'fix%d_b%d_' % (fix, b)
            assert bool(g[tag + 'none']) == (r is None)
            if r is not None:
                for name, t in zip(('iou', 'yx_min', 'yx_max', 'cls', 'score'), r):
                    np.testing.assert_array_equal(t, g[tag + name])


def test_oracle_softmax_close_to_torch(golden):
    d = golden('decode_voc')
    p = odet.softmax(d['logits'])
    np.testing.assert_allclose(p, torch.softmax(torch.from_numpy(d['logits']), -1).numpy(), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize('onehot', [False, True])
def test_loss_matches_reference(golden, onehot):
    g = golden('loss')
    tag = 'onehot_' if onehot else 'ce_'
    gen = torch.Generator().manual_seed(11)
    feat = (0.5 * torch.randn(2, 125, 13, 13, generator=gen)).requires_grad_(True)
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    pred = ohead.decode(feat, anchors)
    data = synth.norm_data(synth.labels(2, 416, 20, seed=2, onehot=onehot), 416, 416, 13, 13)
    loss, debug = oloss.loss(anchors, data, pred, 0.6)
    oloss.total(loss).backward()
    for k in ('foreground', 'background', 'center', 'size', 'cls'):
        np.testing.assert_allclose(loss[k].item(), g[tag + k], rtol=1e-5)
    np.testing.assert_array_equal(debug['positive'].numpy(), g[tag + 'positive'].astype(bool))
    np.testing.assert_array_equal(debug['negative'].numpy(), g[tag + 'negative'].astype(bool))
    np.testing.assert_array_equal(debug['iou'].numpy(), g[tag + 'best_iou'])
    np.testing.assert_allclose(feat.grad.numpy(), g[tag + 'grad'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('onehot', [False, True])
@pytest.mark.parametrize('case', synth.EDGE_CASES, ids=[c[0] for c in synth.EDGE_CASES])
def test_loss_edge_cases_match_reference(golden, case, onehot):
    """Empty image, duplicate boxes, cell collisions, border cells, degenerate boxes (oracle/make_golden_loss_edge.py)."""
    name, S, rows = case
    g = golden('loss_edge')
    tag = '%s_%s_' % (name, 'onehot' if onehot else 'ce')
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    feat = synth.edge_feature(5, 5, 20, rows, rows).requires_grad_(True)
    pred = ohead.decode(feat, anchors)
    data = synth.norm_data(synth.edge_labels(S, S, 20, onehot), S, S, rows, rows)
    loss, debug = oloss.loss(anchors, data, pred, 0.6)
    oloss.total(loss).backward()
    for k in ('foreground', 'background', 'center', 'size', 'cls'):
        np.testing.assert_allclose(loss[k].item(), g[tag + k], rtol=1e-5)
    np.testing.assert_array_equal(debug['positive'].numpy().astype(bool), g[tag + 'positive'].astype(bool))
    np.testing.assert_array_equal(debug['negative'].numpy().astype(bool), g[tag + 'negative'].astype(bool))
    gr = g[tag + 'grad']
    assert np.isfinite(gr).all()
    np.testing.assert_allclose(feat.grad.numpy(), gr, rtol=1e-4, atol=1e-6 * np.abs(gr).max())
