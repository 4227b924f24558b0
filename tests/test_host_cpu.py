"""Host (CPU-tensor) entry points of the PRODUCT library — y2_nms_host / y2_iou_matrix_host / y2_iou_pair_host — behind
utils.postprocess.nms and utils.iou.torch, as the reference's summary worker (train.py:209) and its own IoU unit tests
(utils/iou/torch.py:64-113, 164-213, 240-289) use them: checked against the reference-generated keep lists
(tests/golden/nms.npz), the reference's known-answer IoU cases and the numpy oracle (bit-exact)."""
import numpy as np
import pytest
import torch

import utils.iou.torch as iou
import utils.postprocess as post
from oracle import iou as oiou
from oracle import nms as onms
from oracle import synth

from test_oracle import BIG, KNOWN, QUAD, split


@pytest.mark.parametrize('n', [0, 1, 2, 50, 200, 845, 2000])
def test_nms_host_matches_reference_fixture(golden, n):
    g = golden('nms')
    score, mn, mx = synth.nms_boxes(n)
    for ov, tag in ((0.45, 'ov45'), (0.5, 'ov50')):
        keep = post.nms(torch.from_numpy(score), torch.from_numpy(mn), torch.from_numpy(mx), ov)
        assert isinstance(keep, list)
        np.testing.assert_array_equal(np.asarray(keep, np.int64), g['n%d_%s' % (n, tag)])


def test_nms_host_ties_limit_and_nan():
    rng = np.random.RandomState(5)
    n = 700
    c = rng.uniform(0, 13, (n, 2)).astype(np.float32)
    s = rng.uniform(0.5, 4, (n, 2)).astype(np.float32)
    mn, mx = c - s / 2, c + s / 2
    score = rng.randint(0, 40, n).astype(np.float32) / 40        # many ties: lower index first
    for limit in (1, 7, 64, 65, 200, 1024):
        keep = post.nms(torch.from_numpy(score), torch.from_numpy(mn), torch.from_numpy(mx), 0.45, limit)
        assert keep == onms.nms(score, mn, mx, 0.45, limit)
    # NaN scores rank last (after every number, lower index first): the list is still a valid NMS of the finite scores
    sc = score.copy()
    sc[[3, 100, 699]] = np.nan
    keep = post.nms(torch.from_numpy(sc), torch.from_numpy(mn), torch.from_numpy(mx), 0.45, 1024)
    ref = onms.nms(np.where(np.isnan(sc), -np.inf, sc).astype(np.float32), mn, mx, 0.45, 1024)
    assert keep == ref and len(set(keep)) == len(keep)


@pytest.mark.parametrize('b1,b2,ans', KNOWN)
def test_iou_host_known_answers(b1, b2, ans):
    """utils/iou/torch.py:79-113 (TestIouMatrix), :179-213 (TestBatchIouMatrix) on CPU tensors through the product."""
    t = torch.from_numpy
    mn1, mx1 = split(b1)
    mn2, mx2 = split(b2)
    ans = np.array(ans, np.float32)
    np.testing.assert_almost_equal(iou.iou_matrix(t(mn1), t(mx1), t(mn2), t(mx2)).numpy(), ans)
    rng = np.random.RandomState(0)
    p1 = [rng.permutation(len(b1)) for _ in range(3)]
    p2 = [rng.permutation(len(b2)) for _ in range(3)]
    m = iou.batch_iou_matrix(t(np.stack([mn1[p] for p in p1])), t(np.stack([mx1[p] for p in p1])),
                             t(np.stack([mn2[p] for p in p2])), t(np.stack([mx2[p] for p in p2]))).numpy()
    for b in range(3):
        np.testing.assert_almost_equal(m[b], ans[p1[b]][:, p2[b]])


def test_iou_pair_host_known_answers():
    """utils/iou/torch.py:255-289 (TestBatchIouPair)."""
    t = torch.from_numpy
    mn1, mx1 = split(BIG * 2)
    mn2, mx2 = split(QUAD)
    m = iou.batch_iou_pair(t(mn1[None]), t(mx1[None]), t(mn2[None]), t(mx2[None])).numpy()
    np.testing.assert_almost_equal(m[0], np.array([1 / 7, 4 / 16, 1 / 7, 4 / 16], np.float32))


def test_iou_host_bit_exact_vs_oracle():
    rng = np.random.RandomState(0)
    def boxes(*shape):
        c = rng.uniform(0, 13, shape + (2,)).astype(np.float32)
        s = rng.uniform(0, 6, shape + (2,)).astype(np.float32)
        return c - s / 2, c + s / 2
    mn1, mx1 = boxes(3, 37)
    mn2, mx2 = boxes(3, 11)
    t = torch.from_numpy
    np.testing.assert_array_equal(iou.batch_iou_matrix(t(mn1), t(mx1), t(mn2), t(mx2)).numpy(), oiou.batch_iou_matrix(mn1, mx1, mn2, mx2))
    np.testing.assert_array_equal(iou.iou_matrix(t(mn1[0]), t(mx1[0]), t(mn2[0]), t(mx2[0])).numpy(), oiou.iou_matrix(mn1[0], mx1[0], mn2[0], mx2[0]))
    np.testing.assert_array_equal(iou.intersection_area(t(mn1[0]), t(mx1[0]), t(mn2[0]), t(mx2[0])).numpy(), oiou.intersection_area(mn1[0], mx1[0], mn2[0], mx2[0]))
    a, b = boxes(5, 9)
    c, d = boxes(5, 9)
    np.testing.assert_array_equal(iou.batch_iou_pair(t(a), t(b), t(c), t(d)).numpy(), oiou.batch_iou_pair(a, b, c, d))
    # empty operands (utils/iou/torch.py shapes with N = 0)
    assert iou.iou_matrix(t(mn1[0][:0]), t(mx1[0][:0]), t(mn2[0]), t(mx2[0])).shape == (0, 11)
