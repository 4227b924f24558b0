"""Darknet `.weights` importer + checksum printout.

CPU part: the product importer (convert_darknet_torch.load_darknet_weights) and the oracle restatement
(oracle/darknet_weights.py) against tests/golden/darknet_weights.npz, which holds what the REFERENCE's own
convert_darknet_torch.main() / checksum_torch.main() produced for a seeded synthetic file
(oracle/make_golden_darknet_weights.py regenerates the same bytes from the seed; the fixture records their sha256).
GPU part: a synthetic file imported into the HIP-backed plugin, feature checked against the oracle network fed the same arrays."""
import collections
import configparser
import hashlib

import numpy as np
import pytest
import torch

from oracle import darknet_weights as odw
from oracle import make_golden_darknet_weights as gen
from oracle import synth


def fixture(golden):
    g = golden('darknet_weights')
    shapes = collections.OrderedDict((k, tuple(int(x) for x in v.split(',')) if v else ()) for k, v in zip(g['shape_keys'].tolist(), g['shape_vals'].tolist()))
    data = gen.synthetic_file(shapes)
    assert hashlib.sha256(data).hexdigest() == str(g['file_sha'])       # same bytes the reference consumed
    return g, shapes, data


def plugin(ratio):
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    return model.yolo2.Darknet(model.ConfigChannels(cfg), anchors, gen.NUM_CLS, ratio=ratio), anchors


def test_oracle_importer_matches_reference_output(golden):
    g, shapes, data = fixture(golden)
    mine, header, remaining = odw.read_weights(data, shapes, 5)
    assert header == gen.HEADER and remaining == gen.TRAILING
    assert list(mine.keys()) == g['keys'].tolist()
    for k in mine:
        np.testing.assert_array_equal(mine[k], g['sd/' + k])


def test_product_importer_matches_reference_output(golden, tmp_path):
    import convert_darknet_torch as conv
    g, shapes, data = fixture(golden)
    dnn, anchors = plugin(gen.RATIO)
    sd = dnn.state_dict()
    # the plugin keeps the reference's keys, shapes and ORDER (what the importer's walk depends on)
    assert [k for k in sd if not k.endswith('num_batches_tracked')] == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    path = tmp_path / 'synthetic.weights'
    path.write_bytes(data)
    lines = []
    converted, info = conv.load_darknet_weights(str(path), sd, len(anchors), log=lines.append)
    assert (info['major'], info['minor'], info['revision'], info['seen']) == gen.HEADER
    assert info['remaining'] == gen.TRAILING and info['assigned'] == sum(int(np.prod(s)) for s in shapes.values())
    assert list(converted.keys()) == g['keys'].tolist()
    for k, v in converted.items():
        np.testing.assert_array_equal(v.numpy(), g['sd/' + k])
    assert len(lines) == len(converted) and lines[-1].endswith('remaining=%d' % gen.TRAILING)
    # loads into the plugin (only the torch >= 0.4 counters are absent from a converted checkpoint)
    res = dnn.load_state_dict(converted, strict=False)
    assert not res.unexpected_keys and all(k.endswith('num_batches_tracked') for k in res.missing_keys)
    # checkpoint layout of utils.train.Saver
    out = conv.save_checkpoint(converted, str(tmp_path / 'model'))
    back = torch.load(out)
    assert list(back.keys()) == list(converted.keys())
    # truncated file: loud failure, not a silent partial import
    (tmp_path / 'short.weights').write_bytes(data[:1000])
    with pytest.raises(ValueError, match='ends inside'):
        conv.load_darknet_weights(str(tmp_path / 'short.weights'), sd, len(anchors))


def test_head_row_permutation_is_the_references():
    """x,y,w,h,obj,cls -> iou,y,x,h,w,cls per anchor (convert_darknet_torch.py:37-59), on labelled rows."""
    import convert_darknet_torch as conv
    A, C = 5, 3
    rows = torch.arange(A * (5 + C), dtype=torch.float32)
    got = conv.transpose_bias(rows, A).view(A, -1)
    for a in range(A):
        b = a * (5 + C)
        assert got[a].tolist() == [b + 4, b + 1, b + 0, b + 3, b + 2, b + 5, b + 6, b + 7]
    w = rows.view(-1, 1, 1, 1).repeat(1, 2, 1, 1)
    assert torch.equal(conv.transpose_weight(w, A)[:, 0, 0, 0], got.reshape(-1))
    np.testing.assert_array_equal(odw.permute_head_rows(rows.numpy(), A), got.reshape(-1).numpy())
    with pytest.raises(ValueError):
        conv.transpose_bias(torch.zeros(23), 5)


def test_checksum_rows_match_reference_printout(golden):
    """Parameter rows of checksum_torch (abs-mean text + md5) are bit-identical to the reference's printout for the same
    checkpoint; computed here from the product's row() on the converted arrays (no GPU needed for the parameter rows)."""
    import checksum_torch as chk
    g, shapes, data = fixture(golden)
    ref_rows = g['checksum_rows'].tolist()
    mine, _, _ = odw.read_weights(data, shapes, 5)
    ours = [chk.row(k, v) for k, v in mine.items()]
    want = [r for r in ref_rows if r.split('\t')[0] in mine]
    assert sorted(ours) == sorted(want) and len(want) == len(mine)
    assert [odw.checksum_row(k, v) for k, v in mine.items()] == ours
    # (the `tensor` / `output` rows of the printout depend on the RNG stream after the reference's model constructor and on
    # torch-CPU arithmetic; the GPU test below checks those two rows against the oracle network instead)


@pytest.mark.gpu
def test_imported_weights_run_on_the_hip_plugin_and_match_oracle(tmp_path):
    """Synthetic .weights -> importer -> HIP plugin (eval) vs the oracle network on the same arrays (fp64), and the checksum
    printout of the plugin against rows computed from the oracle's arrays / fp64 output."""
    import checksum_torch as chk
    import convert_darknet_torch as conv
    from oracle import darknet as odark
    ratio = 1.0 / 8
    dnn, anchors = plugin(ratio)
    sd = dnn.state_dict()
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in sd.items() if not k.endswith('num_batches_tracked'))
    rng = np.random.RandomState(7)
    arrays = []
    for key, shape in odw.file_order(shapes):
        fan_in = int(np.prod(shape[1:])) if len(shape) == 4 else 1
        a = rng.standard_normal(shape).astype(np.float32) * (np.sqrt(2.0 / fan_in) if len(shape) == 4 else 0.1)
        if key.endswith('running_var') or key.endswith('bn.weight'):
            a = np.abs(a) + 0.5
        if key.startswith('layers3.1'):
            a = a / 8
        arrays.append(a)
    data = odw.write_weights(arrays, (0, 2, 0, 99))
    (tmp_path / 'w.weights').write_bytes(data)
    converted, info = conv.load_darknet_weights(str(tmp_path / 'w.weights'), sd, len(anchors))
    assert info['remaining'] == 0
    oracle_sd, _, _ = odw.read_weights(data, shapes, len(anchors))
    dnn.load_state_dict(converted, strict=False)
    dnn.cuda().eval()
    torch.manual_seed(0)
    tensor = torch.randn(1, 3, 416, 416)
    rows = chk.checksum_rows(dnn, tensor)
    with torch.no_grad():
        ref = odark.forward(tensor.double(), {k: torch.from_numpy(v).double() for k, v in oracle_sd.items()})
        got = dnn(tensor.cuda())
    rms = ref.pow(2).mean().sqrt().item()
    assert (got.double().cpu() - ref).abs().max().item() <= 2e-5 * rms
    want = [odw.checksum_row(k, v) for k, v in oracle_sd.items()] + [odw.checksum_row('tensor', tensor.numpy()), odw.checksum_row('output', ref.float().numpy())]
    assert chk.compare_rows(rows, want) == []
    # a corrupted checkpoint is caught by the fingerprint
    bad = list(want)
    f = bad[3].split('\t')
    f[3] = '0' * 32
    bad[3] = '\t'.join(f)
    assert [k for k, _ in chk.compare_rows(rows, bad)] == [f[0]]
