"""RCCL path of the data-parallel wrapper on the single visible MI355X (world_size 1, backend nccl): the Darknet
backward announces gradients layer by layer (grad_ready_hook), buckets are all-reduced through RCCL and written back;
with one rank the result must equal the unwrapped gradients (up to the run-to-run summation-order noise of the split-K
weight-gradient atomics)."""
import configparser
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import darknet as odark
from oracle import loss as oloss
from oracle import synth
from oracle.make_golden import NARROW

pytestmark = pytest.mark.gpu


def build(sd):
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, 20)
    dnn.load_state_dict(sd, strict=False)
    return model.Inference(cfg, dnn, anchors).cuda().train(), anchors


def test_dp_wrapper_rccl_world1_equals_plain_backward():
    import model
    import train
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    try:
        widths = dict(NARROW)
        widths['layers1.5'] = 8
        sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0)
        x = synth.images(2, 96, seed=1).cuda()
        data = synth.norm_data(synth.labels(2, 96, 20, seed=2), 96, 96, 3, 3)
        grads = []
        for wrap in (False, True):
            inf, anchors = build(sd)
            m = train.DataParallelRCCL(inf, bucket_bytes=4096) if wrap else inf
            if wrap:
                assert inf.dnn.grad_ready_hook is not None
            pred = model._inference(m, x)
            loss, _ = model.loss(anchors, data, pred, 0.6)
            sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
            torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in inf.named_parameters()})
        for k in grads[0]:
            a, b = grads[0][k], grads[1][k]
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-9, k
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_between_replayed_graph_segments_world1():
    """train.iterate under the wrapper on RCCL itself (world size 1, the one GPU there is): from the 4th step on the step is a chain of
    hipGraph segments with eager RCCL all-reduces of the gradient buckets between them, on the same stream.  Eight steps at learning rate 0
    must reproduce the un-wrapped autograd step (averaging over one rank is the identity) - graph replays, RCCL launches and the
    wrapper's waits in their real interplay."""
    import train
    import utils
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port_early()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.pop('Y2_DIST_BACKEND', None)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    try:
        widths = dict(NARROW)
        widths['layers1.5'] = 8
        sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0)
        data = []
        for i, nmax in enumerate((5, 8, 3)):
            d = {k: v.cuda() for k, v in synth.labels(3, 96, 20, nmax=nmax, seed=20 + i).items()}
            d['tensor'] = synth.images(3, 96, seed=30 + i).cuda()
            data.append(d)
        runs = []
        for wrap in (False, True):
            inf, anchors = build(sd)
            train.PLAN = wrap                       # the witness: plain autograd, no wrapper
            try:
                m = train.DataParallelRCCL(inf, bucket_bytes=4096) if wrap else inf
                opt = utils.optim.SGD(m.parameters(), 0.0, momentum=0.9)
                rows = []
                for i in range(8):
                    r = train.iterate(m, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
                    rows.append(([float(r['loss'][k].detach()) for k in r['loss']], {k: p.grad.detach().clone() for k, p in inf.dnn.named_parameters()}))
                torch.cuda.synchronize()
                runs.append(rows)
                if wrap:
                    runner = inf.__dict__['_y2_step_runner']
                    plan = next(iter(runner.plans.values()))
                    kinds = [op[0] for op in plan.ops]
                    assert runner.captures == 1 and kinds.count('graph') >= 3 and 'buckets' in kinds and 'npos' not in kinds, kinds
            finally:
                train.PLAN = True
        for (la, ga), (lb, gb) in zip(*runs):
            for a, b in zip(la, lb):
                assert abs(a - b) <= 2e-5 * abs(a) + 1e-12
            for k in ga:
                assert (ga[k] - gb[k]).abs().max().item() <= 1e-3 * ga[k].pow(2).mean().sqrt().item() + 1e-12, k
    finally:
        dist.destroy_process_group()


def _free_port_early():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ------------------------------------------------------------------------------------------------ world_size 2 on the real path
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


DP_S, DP_B, DP_LABEL_SEED = 96, 4, 2


def _dp_inputs():
    widths = dict(NARROW)
    widths['layers1.5'] = 8
    sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0)
    x = synth.images(DP_B, DP_S, seed=1)
    data = synth.norm_data(synth.labels(DP_B, DP_S, 20, seed=DP_LABEL_SEED), DP_S, DP_S, DP_S // 32, DP_S // 32)
    return sd, x, data


def _dp_rank(rank, world, port, tmp, sync):
    """One data-parallel rank: Darknet (narrow) + model.loss on its shard through train.ensure_model -> DataParallelRCCL.
    On a box with one GPU both ranks share cuda:0 and the process group is gloo with host-staged buffers; with >= 2 GPUs visible each
    rank takes its own GPU and the group is RCCL.  Everything else - the grad_ready_hook bucket protocol, the positive-count
    all-reduce, the HIP kernels - is the product path either way."""
    import sys
    from conftest import APP, ROOT
    for p in (ROOT, APP):
        if p not in sys.path:
            sys.path.insert(0, p)
    rccl = torch.cuda.device_count() >= world and os.environ.get('Y2_TEST_DP_BACKEND', 'auto') != 'gloo'
    if rccl:       # a multi-GPU node: the real thing, one GPU per rank over RCCL
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        os.environ.pop('Y2_DIST_BACKEND', None)
    else:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', Y2_DIST_BACKEND='gloo')
    import model
    import train
    from model import train_graph
    train_graph.SYNC_POSITIVES = sync
    assert train.init_distributed() == world
    sd, x, data = _dp_inputs()
    if rank != 0:       # replicas must end up with rank 0's weights (the wrapper broadcasts them)
        sd = {k: (v + 0.01 if v.is_floating_point() else v) for k, v in sd.items()}
    inf, anchors = build(sd)
    m = train.ensure_model(inf)
    assert isinstance(m, train.DataParallelRCCL) and inf.dnn.grad_ready_hook is not None
    per = DP_B // world
    sl = slice(rank * per, (rank + 1) * per)
    pred = model._inference(m, x[sl].cuda())
    loss, debug = model.loss(anchors, {k: v[sl] for k, v in data.items()}, pred, 0.6)
    sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
    torch.cuda.synchronize()
    torch.save({'loss': {k: v.item() for k, v in loss.items()}, 'positives': int(debug['positive'].sum().item()),
                'grads': {k: p.grad.cpu() for k, p in inf.dnn.named_parameters()},
                'buffers': {k: b.cpu() for k, b in inf.dnn.named_buffers()}}, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def _oracle_concatenated_batch(world, dtype=torch.float64):
    """The reference's semantics for the global batch (train.py:65-71, 296-309; model/__init__.py:159-166): BatchNorm statistics
    per replica shard, ONE loss over the concatenated batch (cnt = B_total*cells*A, cls = mean over ALL positives), fp64 autograd."""
    from oracle import head as ohead
    sd, x, data = _dp_inputs()
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    sd64 = {k: v.to(dtype).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    per = DP_B // world
    stats0 = {}
    feats = [odark.forward(x[r * per:(r + 1) * per].to(dtype), sd64, training=True, stats=(stats0 if r == 0 else {})) for r in range(world)]
    lo, dbg = oloss.loss(anchors.to(dtype), {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in data.items()},
                         ohead.decode(torch.cat(feats, 0), anchors.to(dtype)), 0.6)
    oloss.total(lo).backward()
    return sd64, lo, dbg, stats0


def _rel(got, ref):
    rms = ref.double().pow(2).mean().sqrt().item()
    return (got.double() - ref.double()).abs().max().item() / max(rms, 1e-30)


@pytest.mark.timeout(600)
def test_dp_world2_darknet_region_loss_equals_concatenated_batch(tmp_path):
    """two ranks x B/2 on Darknet + region loss == one process on the concatenated batch (per-shard BN), including
    the positive-count all-reduce that makes the mean-over-positives cls term global (model/__init__.py:162)."""
    import torch.multiprocessing as mp
    world = 2
    sd64, lo, dbg, stats0 = _oracle_concatenated_batch(world)
    sd32 = _oracle_concatenated_batch(world, torch.float32)[0]        # the same step in fp32 on the oracle: the noise floor of fp32 arithmetic
    per = DP_B // world
    pos = dbg['positive'].view(DP_B, -1).sum(1)
    assert pos[:per].sum().item() != pos[per:].sum().item(), 'the shards must hold different numbers of positives for this test to bite'
    out = {}
    for sync in (True, False):
        d = tmp_path / ('sync%d' % sync)
        d.mkdir()
        mp.spawn(_dp_rank, args=(world, _free_port(), str(d), sync), nprocs=world, join=True)
        out[sync] = [torch.load(str(d / ('rank%d.pt' % r)), weights_only=False) for r in range(world)]
    r0, r1 = out[True]
    assert r0['positives'] + r1['positives'] == int(pos.sum().item()) and r0['positives'] != r1['positives']
    # averaged gradients are identical on both ranks and equal the single-process gradient on the concatenated batch
    for k, v in sd64.items():
        if not v.requires_grad:
            continue
        assert torch.equal(r0['grads'][k], r1['grads'][k]), k
        e = _rel(r0['grads'][k], v.grad)
        assert e <= max(2e-4, 2.5 * _rel(sd32[k].grad, v.grad)), (k, e)          # the stated gradient tolerance, as in the single-process training-step test
    # loss terms: every rank divides by its LOCAL cnt, so the mean over ranks is the global term; cls uses the GLOBAL positive count
    for k in lo:
        got = 0.5 * (r0['loss'][k] + r1['loss'][k])
        assert abs(got - lo[k].item()) <= 1e-4 * abs(lo[k].item()), (k, got, lo[k].item())
    # rank 0's running statistics are those of shard 0 (only replica 0's buffers persist in nn.DataParallel)
    for prefix, (rm, rv) in stats0.items():
        assert _rel(r0['buffers'][prefix + '.bn.running_mean'], rm) <= 1e-4, prefix
        assert _rel(r0['buffers'][prefix + '.bn.running_var'], rv) <= 1e-4, prefix
    # without the positive-count all-reduce the cls term (and with it the head gradient) is visibly wrong
    n0, n1 = out[False]
    wrong = 0.5 * (n0['loss']['cls'] + n1['loss']['cls'])
    assert abs(wrong - lo['cls'].item()) > 1e-3 * abs(lo['cls'].item())
    for k in ('foreground', 'background', 'center', 'size'):
        assert abs(0.5 * (n0['loss'][k] + n1['loss'][k]) - lo[k].item()) <= 1e-4 * abs(lo[k].item()), k
    head = 'layers3.1.conv.weight'
    assert _rel(n0['grads'][head], sd64[head].grad) > 10 * _rel(r0['grads'][head], sd64[head].grad)


def test_wrapped_model_survives_cpu_cuda_round_trip():
    """train.py:423-432: Train.eval moves the (wrapped) inference module to the CPU and back between training steps.  The plan /
    packed-weight caches are keyed on data pointers and the wrapper's flat buckets live on the old device: after the round trip a
    training step and an eval forward must still be right (fresh model on the same weights as the witness)."""
    import model
    import train
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.pop('Y2_DIST_BACKEND', None)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    try:
        widths = dict(NARROW)
        widths['layers1.5'] = 8
        sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0)
        x = synth.images(2, 96, seed=1).cuda()
        data = synth.norm_data(synth.labels(2, 96, 20, seed=2), 96, 96, 3, 3)

        def step(m, anchors):
            for p in m.parameters():
                p.grad = None
            pred = model._inference(m, x)
            loss, _ = model.loss(anchors, data, pred, 0.6)
            sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
            torch.cuda.synchronize()

        inf, anchors = build(sd)
        m = train.DataParallelRCCL(inf, bucket_bytes=4096)
        step(m, anchors)                                   # buckets and caches now live on cuda:0
        inf.eval()
        with torch.no_grad():
            before = model._inference(m, x)['feature'].clone()
        m.cpu()                                            # train.py:424
        assert all(not p.is_cuda for p in inf.parameters())
        m.cuda()                                           # train.py:432
        with torch.no_grad():
            after = model._inference(m, x)['feature']
        assert torch.equal(before, after)                  # same weights, re-packed from the new allocations
        inf.train()
        step(m, anchors)
        witness, anchors2 = build({k: v.clone() for k, v in inf.dnn.state_dict().items()})
        # train-mode gradients do not depend on the running statistics, so a fresh model on the same weights is an exact witness
        step(witness, anchors2)
        for (k, a), (_, b) in zip(inf.named_parameters(), witness.named_parameters()):
            assert (a.grad - b.grad).abs().max().item() <= 1e-5 * b.grad.abs().max().item() + 1e-9, k
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_bench_two_ranks_share_the_one_gpu():
    """bench.py's N > 1 code path end to end on the real kernels: `--gpus 2` launches the ranks itself; the test rig puts both on
    cuda:0 (Y2_BENCH_DEVICE) with a gloo group standing in for RCCL (Y2_DIST_BACKEND).  The numbers mean nothing (two ranks share one
    GPU, gradients staged through the host); the protocol is what is checked: rendezvous, ranks_seen, detect replicas, the
    no-wrapper training leg beside the data-parallel one, ONE JSON line with the train headline."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(Y2_DIST_BACKEND='gloo', Y2_BENCH_DEVICE='0')
    import tempfile
    tables = os.path.join(tempfile.mkdtemp(), 'full.json')
    # (three small sizes for the multi-scale leg: two ranks' plans for ten sizes do not fit ONE GPU's memory - a property of this rig)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--train-steps', '3', '--cpu-sample', '0', '--no-resnet',
                          '--ms-sizes', '320,352,384', '--ms-maintain', '4', '--tables', tables], capture_output=True, text=True, timeout=850, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) <= 6000                    # ONE compact line (the driver parses the last line of stdout) ...
    rec = json.loads(lines[0])
    full = json.load(open(tables))                                      # ... and the long form in the tables file
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and rec['headline'] == 'train' and rec['scaling'] == 'weak'
    assert rec['value'] == full['train']['images_per_sec'] > 0 and full['train']['global_batch'] == 128 and rec['config']['global_batch'] == 128
    assert full['train']['single_gpu_images_per_sec'] > 0 and 'dp2' in full['train']['parallelism'] and 'dp2' in rec['config']['parallelism']
    assert full['detect']['images_per_sec'] > 0 and 'replicas x2' in full['detect']['parallelism']
    assert 'roofline' not in rec and 'cpu_baseline' not in rec          # N = 1 only
    sm = rec['summary']                                                 # the legs' scalars, once
    assert sm['train_images_per_sec'] == rec['value'] and sm['train_single_gpu_images_per_sec'] > 0 and sm['detect_images_per_sec'] > 0 and sm['multiscale_images_per_sec'] > 0
    assert 'train_dp_exposed_comm_ms_per_step' in sm
    assert full['train']['autotune_choices_synced'] and full['train']['autotune_choices_synced'] > 10      # rank 1 adopted rank 0's algorithm table
