"""Data-parallel wrapper (train.ensure_model / DataParallelRCCL) on CPU with the gloo backend, world_size 2:
averaged gradients must equal single-process gradients on the concatenated batch, for both the late path
(post-accumulate hooks) and the early path (a module announcing gradients from inside its backward, like the
Darknet training graph does layer by layer)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import APP, ROOT


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class EarlyLinear(nn.Module):
    """A layer whose backward announces its parameter gradients through grad_ready_hook (the Darknet protocol)."""

    def __init__(self, i, o):
        nn.Module.__init__(self)
        self.weight = nn.Parameter(torch.randn(o, i) * 0.3)
        self.bias = nn.Parameter(torch.zeros(o))
        self.grad_ready_hook = None
        self.grad_buffer_hook = None       # the wrapper's offer to write gradients straight into its flat buckets
        self.in_place = 0                  # how many gradients this module wrote into a bucket slice

    def forward(self, x):
        mod = self

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w, b):
                ctx.save_for_backward(x, w)
                return x @ w.t() + b

            @staticmethod
            def backward(ctx, g):
                x, w = ctx.saved_tensors
                gw, gb = g.t() @ x, g.sum(0)
                if mod.grad_buffer_hook is not None:      # the Darknet backward's protocol: ask for the destination first
                    slot = mod.grad_buffer_hook(mod.weight)
                    if slot is not None:
                        assert slot.shape == gw.shape
                        gw = slot.copy_(gw)
                        mod.in_place += 1
                if mod.grad_ready_hook is not None:
                    mod.grad_ready_hook(mod.bias, gb)
                    mod.grad_ready_hook(mod.weight, gw)
                return g @ w, gw, gb
        return Fn.apply(x, self.weight, self.bias)


def make_model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 16), nn.Tanh(), EarlyLinear(16, 12), nn.Tanh(), nn.Linear(12, 3))


def worker(rank, world, port, tmp):
    for p in (ROOT, APP):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import train
    assert train.init_distributed() == world
    torch.manual_seed(100 + rank)          # different initial weights per rank: the wrapper must broadcast rank 0's
    m = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), EarlyLinear(16, 12), nn.Tanh(), nn.Linear(12, 3))
    ref = make_model()
    if rank == 0:
        m.load_state_dict(ref.state_dict())
    dp = train.DataParallelRCCL(m, bucket_bytes=256)   # tiny buckets: several all-reduces per step
    for a, b in zip(m.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)
    for step in range(2):                  # twice: state must reset between steps
        for p in dp.parameters():
            p.grad = None
        ((dp(x[shard]) - y[shard]) ** 2).mean().backward()
        for p in ref.parameters():
            p.grad = None
        ((ref(x) - y) ** 2).mean().backward()
        for (n, a), b in zip(m.named_parameters(), ref.parameters()):
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), (n, step)
    # gradient accumulation (no zero_grad between two backward passes) through both hook paths: grad = avg(step 1) + avg(step 2)
    for p in dp.parameters():
        p.grad = None
    for p in ref.parameters():
        p.grad = None
    for rep in range(2):
        sh = slice(rank * 4, rank * 4 + 4) if rep == 0 else slice(4 - rank * 4, 8 - rank * 4)
        ((dp(x[sh]) - y[sh]) ** 2).mean().backward()
        ((ref(x) - y) ** 2).mean().backward()
    for (n, a), b in zip(m.named_parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), ('accumulate', n)
    assert m[2].in_place >= 2, m[2].in_place            # the zero_grad'ed steps wrote the early layer's weight gradient in place
    # Train.iterate's ordering (train.py:344-351): forward FIRST, then optimizer.zero_grad(), then backward - last step's gradients are
    # still set while forward() runs; who accumulates must be decided at backward time or the in-place bucket path never runs
    before = m[2].in_place
    for step in range(2):
        out = dp(x[shard])
        for p in dp.parameters():
            p.grad = None
        ((out - y[shard]) ** 2).mean().backward()
        for p in ref.parameters():
            p.grad = None
        ((ref(x) - y) ** 2).mean().backward()
        for (n, a), b in zip(m.named_parameters(), ref.parameters()):
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), ('iterate order', n, step)
        assert m[2].weight.grad.data_ptr() == dp._flat[dp._where[id(m[2].weight)][0]][dp._where[id(m[2].weight)][1]:].data_ptr()      # the gradient IS the bucket slice
    assert m[2].in_place == before + 2, (m[2].in_place, before)
    # two backward passes WITHOUT a forward in between (two outputs of one step): the first pass's averaged gradients are bucket
    # slices; the second pass must not overwrite them in place: grad = avg(pass 1) + avg(pass 2)
    for p in dp.parameters():
        p.grad = None
    for p in ref.parameters():
        p.grad = None
    o1 = dp(x[shard])
    l1, l2 = ((o1 - y[shard]) ** 2).mean(), (o1 ** 2).mean()
    l1.backward(retain_graph=True)
    l2.backward()
    r1 = ref(x)
    (((r1 - y) ** 2).mean() + (r1 ** 2).mean()).backward()
    for (n, a), b in zip(m.named_parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), ('two backwards', n)
    # a step issued as captured graph segments (model.train_graph.StepPlan) drives the wrapper through graph_begin / graph_slot / graph_launch /
    # graph_finish instead of the autograd hooks: same collectives, same order, same payloads - rank 0 on that protocol and rank 1 on the hooks
    # must meet (a rank replaying graphs and a rank still warming up on another input size do exactly this)
    for step in range(2):
        for p in dp.parameters():
            p.grad = None
        if rank == 0:
            hooks = (m[2].grad_ready_hook, m[2].grad_buffer_hook)
            m[2].grad_ready_hook = m[2].grad_buffer_hook = None          # (a StepPlan takes the module's hook slots over for its own pass, too)
            local = torch.autograd.grad(((m(x[shard]) - y[shard]) ** 2).mean(), list(m.parameters()))
            m[2].grad_ready_hook, m[2].grad_buffer_hook = hooks
            dp.graph_begin()
            nb = len(dp._buckets)
            done = 0
            for bi, bucket in enumerate(dp._buckets):           # gradients land in their slices; buckets go out as they complete
                for p in bucket:
                    g = local[[id(q) for q in m.parameters()].index(id(p))]
                    dp.graph_slot(p).copy_(g)
                if bi < nb - 1:
                    dp.graph_launch(bi, bi + 1)
                    done = bi + 1
            dp.graph_launch(done, nb)
            dp.graph_finish()
            for p in m.parameters():
                p.grad = dp.graph_slot(p)
        else:
            ((dp(x[shard]) - y[shard]) ** 2).mean().backward()
        for p in ref.parameters():
            p.grad = None
        ((ref(x) - y) ** 2).mean().backward()
        for (n, a), b in zip(m.named_parameters(), ref.parameters()):
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), ('graph protocol vs hooks', n, step, rank)
    # the reducer of the region loss travels with the wrapper's outputs
    from model import train_graph
    out = dp(x[shard])
    red = getattr(out, train_graph.DP_TAG)
    cnt = torch.tensor([float(rank + 1)])
    assert red(cnt) and cnt.item() == 3.0
    assert getattr(m(x[shard]), train_graph.DP_TAG, None) is None       # the bare module's outputs carry none
    # replicated inference: every rank adopts rank 0's measured algorithm table
    import _hip
    _hip._TUNE.clear()
    _hip._TUNE[(32, 13, 13, 1024, 'cpu')] = [rank + 1, 5]
    epoch = _hip.tune_epoch()
    assert train.sync_tune(torch.device('cpu')) == 1
    assert _hip._TUNE[(32, 13, 13, 1024, 'cpu')] == [1, 5]
    assert (_hip.tune_epoch() != epoch) == (rank != 0)          # plans built on the old choices are invalidated where the table changed
    # the wrapper's own exchange is triggered by its call count alone (identical on every rank whatever shapes the ranks' loaders drew) and
    # MERGES: rank 0's entries win, an entry only this rank has stays
    _hip._TUNE.clear()
    _hip._TUNE[(64, 10, 10, 512, 'cpu')] = [1, rank]
    _hip._TUNE[('mine', rank, 'cpu')] = [2, 0]
    dp2 = train.DataParallelRCCL(nn.Linear(3, 3))
    for call in range(1, 5):
        dp2(torch.randn(2 + rank * call, 3)).sum().backward()   # a different input shape per rank and call; a step is a BACKWARD that all-reduces
        assert (dp2.tune_synced is not None) == (call >= 4)
    assert _hip._TUNE[(64, 10, 10, 512, 'cpu')] == [1, 0] and _hip._TUNE[('mine', rank, 'cpu')] == [2, 0]
    assert (('mine', 0, 'cpu') in _hip._TUNE) and len(_hip._TUNE) == (2 if rank == 0 else 3)
    # only TRAINING steps advance that schedule: rank 0 alone runs evaluation / no_grad forwards through the wrapper (the summary
    # worker, an eval pass) between training steps; were they counted, rank 0 would reach the tune broadcast one training step before rank 1,
    # which would meet it with a gradient all-reduce on the same group - a hang or garbage
    lin = nn.Linear(3, 3)
    dp3 = train.DataParallelRCCL(lin)
    xs = torch.randn(4, 3, generator=torch.Generator().manual_seed(5))
    for call in range(1, 6):
        if rank == 0:
            with torch.no_grad():
                dp3(xs)                                         # rank-local, gradients off
            lin.eval()
            dp3(xs)                                             # rank-local, eval mode
            lin.train()
        for p in lin.parameters():
            p.grad = None
        if call == 3:
            lin.eval()                                          # a differentiable eval()-mode step on EVERY rank (frozen-BatchNorm fine-tuning) is a training step
        dp3(xs[rank * 2:rank * 2 + 2]).pow(2).mean().backward()
        lin.train()
        assert dp3._calls == call, (rank, dp3._calls, call)
        assert (dp3.tune_synced is not None) == (call >= 4), (rank, call)
        red = [p.grad.clone() for p in lin.parameters()]
        # witness on private copies of the weights (a backward through `lin` itself would fire the wrapper's hooks: that IS a step of the wrapper)
        wit = [p.detach().clone().requires_grad_() for p in lin.parameters()]
        torch.nn.functional.linear(xs, wit[0], wit[1]).pow(2).mean().backward()
        for a, w in zip(red, wit):
            assert torch.allclose(a, w.grad, rtol=1e-5, atol=1e-6), ('eval calls between training steps', rank, call)
    if rank == 0:
        open(os.path.join(tmp, 'ok'), 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


class Gated(nn.Module):
    """Two branches + one parameter nobody uses: which branch runs is decided per call (per rank in the test)."""

    def __init__(self):
        nn.Module.__init__(self)
        self.a, self.b, self.unused = nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4)
        self.out = nn.Linear(4, 2)

    def forward(self, x, use_b):
        h = self.a(x)
        if use_b:
            h = h + self.b(x)
        return self.out(torch.tanh(h))


def uneven_worker(rank, world, port, tmp):
    """Robustness of the bucket protocol: a parameter that only ONE rank produced a gradient for, a parameter nobody used, and a
    backward that raised half-way must neither desynchronise the collectives nor poison the next step."""
    for p in (ROOT, APP):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import train
    train.init_distributed()
    torch.manual_seed(3)
    m = Gated()
    dp = train.DataParallelRCCL(m, bucket_bytes=64)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 4, generator=g)
    # (1) `b` runs on rank 1 only
    dp(x[rank * 2:rank * 2 + 2], use_b=(rank == 1)).pow(2).sum().backward()
    ref = Gated()
    ref.load_state_dict(m.state_dict())
    l0 = ref(x[0:2], False).pow(2).sum()
    l1 = ref(x[2:4], True).pow(2).sum()
    ((l0 + l1) / 2).backward()
    for (n, a), b in zip(m.named_parameters(), ref.parameters()):
        if n.startswith('unused'):
            assert a.grad is None, n                     # nobody produced a gradient: stays None on every rank
        else:
            assert a.grad is not None, (n, rank)         # rank 0 receives b's averaged gradient although it never ran b
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), (n, rank)
    # (2) a backward that raises leaves no stale state behind
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, gr):
            raise RuntimeError('boom')
    for q in m.parameters():
        q.grad = None
    try:
        Boom.apply(dp(x[rank * 2:rank * 2 + 2], use_b=False)).sum().backward()
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    for q in m.parameters():
        q.grad = None
    dp(x[rank * 2:rank * 2 + 2], use_b=False).pow(2).sum().backward()
    for q in ref.parameters():
        q.grad = None
    ((ref(x[0:2], False).pow(2).sum() + ref(x[2:4], False).pow(2).sum()) / 2).backward()
    for (n, a), b in zip(m.named_parameters(), ref.parameters()):
        if n.startswith('unused') or n.startswith('b.'):
            assert a.grad is None, n
        else:
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6), (n, rank)
    if rank == 0:
        open(os.path.join(tmp, 'ok'), 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_dp_uneven_gradients_and_failed_backward_gloo_world2(tmp_path):
    port = free_port()
    mp.spawn(uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


@pytest.mark.timeout(180)
def test_dp_wrapper_gloo_world2(tmp_path):
    port = free_port()
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


def test_ensure_model_single_process_is_identity():
    import train
    m = nn.Linear(2, 2)
    assert train.ensure_model(m) is m or not torch.cuda.is_available()


def test_norm_data():
    import train
    d = dict(yx_min=torch.tensor([[[104.0, 208.0]]]), yx_max=torch.tensor([[[416.0, 416.0]]]), cls=torch.tensor([[3]]))
    n = train.norm_data(d, 416, 416, 13, 13)
    assert n['yx_min'].tolist() == [[[3.25, 6.5]]] and n['yx_max'].tolist() == [[[13.0, 13.0]]] and n['cls'] is d['cls']


@pytest.mark.timeout(300)
def test_bench_self_launches_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher must spawn the two ranks itself (torch.distributed.run, 127.0.0.1), rendezvous,
    all-reduce `ranks_seen`, run the DP wrapper and print ONE JSON line on rank 0.  --dry-run swaps the GPU hot path (which has no
    CPU fallback) for a stand-in CPU workload and flags the line as invalid: this test is about the launch protocol only."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1'],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and rec['dry_run'] is True and rec['valid'] is False
    assert rec['headline'] == 'train' and rec['train']['wrapper'] == 'DataParallelRCCL' and rec['steps'] == 3
    # the line of the 1-GPU run names the SAME workload (the driver divides value(N) by value(1)): the batch-64 training step at every N
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--dry-run', '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=120, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    rec1 = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][0])
    assert rec1['n_gpus'] == 1 and rec1['headline'] == rec['headline'] == 'train' and rec1['metric'] == rec['metric']
    assert rec1['config']['workload'] == rec['config']['workload'] and 'configs[2]' in rec['config']['workload'] and 'batch-64/GPU' in rec['config']['workload']
    assert rec['config']['global_batch'] == 2 * rec1['config']['global_batch'] == 128
    # a world size that contradicts --gpus is refused, not silently accepted
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '1'],
                         capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert bad.returncode != 0 and 'WORLD_SIZE=1' in (bad.stderr + bad.stdout)


def test_buckets_follow_the_plugins_backward_order_and_keep_one_dimensional_parameters_apart():
    """The all-reduce of a bucket can start when its LAST gradient exists.  Darknet's backward finishes weights in the order layers3,
    layers2, passthrough, layers1 and hands the BatchNorm / bias gradients of all layers out together at the very end: buckets must
    hold weights in that order and keep every one-dimensional parameter in the last bucket, or no bucket completes before the end."""
    import configparser
    for p in (ROOT, APP):
        if p not in sys.path:
            sys.path.insert(0, p)
    import model
    import model.yolo2
    import train
    from oracle import darknet as odark
    from oracle import synth
    from oracle.make_golden import NARROW
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        cfg = configparser.ConfigParser()
        cfg.read_dict({'batch_norm': {'enable': '1'}})
        anchors = torch.from_numpy(synth.ANCHORS_VOC)
        sd = odark.init_state_dict(5, 20, seed=0, channels=NARROW)
        dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, 20)
        dp = train.DataParallelRCCL(model.Inference(cfg, dnn, anchors), bucket_bytes=2048)
        assert len(dp._buckets) > 4
        assert all(p.dim() <= 1 for p in dp._buckets[-1]) and all(p.dim() > 1 for b in dp._buckets[:-1] for p in b)
        flat = [p for b in dp._buckets[:-1] for p in b]
        want = dnn.backward_param_order()
        assert [id(p) for p in flat] == [id(p) for p in want]
        names = {id(p): n for n, p in dnn.named_parameters()}
        order = [names[id(p)] for p in flat]
        assert order[0] == 'layers3.1.conv.weight' and order.index('passthrough.conv.weight') > order.index('layers2.1.conv.weight')
        assert sorted(id(p) for b in dp._buckets for p in b) == sorted(id(p) for p in dnn.parameters())
    finally:
        dist.destroy_process_group()
