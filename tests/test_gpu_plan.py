"""A training step as a StepPlan (model.train_graph.StepPlan / train.StepRunner): the launch sequence of train.py:344-351 issued
without autograd and, after the warm-up steps, replayed from captured hipGraph segments.  It must be the SAME step as the autograd
path: same losses, same gradients, same running statistics, same parameters after a few optimizer steps - with labels whose box
count changes from step to step (the static label buffers are zero-padded), for every plugin, and under the data-parallel wrapper
(two ranks, also with one rank on the graph path and the other on the autograd path: the collective sequences must match)."""
import configparser
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import darknet as odark
from oracle import loss as oloss
from oracle import synth
from oracle.make_golden import NARROW

pytestmark = pytest.mark.gpu

# What separates two runs of the same step on the GPU: completion-order fp32 / fp64 atomics (split-K weight gradients, BatchNorm sums).
# The comparisons below therefore run with learning rate 0 - every step is the same function of (weights, batch), whichever path issues
# it - and hold losses / predictions to fp32 rounding and gradients to that noise.  (Trajectories of a 27-samples-per-channel BatchNorm
# network are chaotic: one IoU crossing the 0.6 background threshold moves a loss term by 1 %; comparing them says nothing about the path.)
# That replayed graphs really train is checked separately against an autograd witness on the weights they produced.
STEP_TOL = 2e-5


def dev():
    return torch.device('cuda', 0)


def rel(got, ref):
    rms = ref.double().pow(2).mean().sqrt().item()
    return (got.double() - ref.double()).abs().max().item() / max(rms, 1e-30)


def darknet_sd():
    widths = dict(NARROW)
    widths['layers1.5'] = 8
    return odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0)


def build(kind, sd=None, C=20):
    import model
    import model.resnet
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}, 'model': {'pretrained': '0'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    if kind == 'darknet':
        sd = darknet_sd() if sd is None else sd
        dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, C)
    elif kind == 'tiny':
        sd = odark.init_tiny_state_dict(5, C, seed=0, div=4, head_scale=0.25)
        dnn = model.yolo2.Tiny(model.ConfigChannels(cfg, sd), anchors, C)
    else:
        from oracle import resnet as ores
        sd = ores.init_state_dict(kind, 5, C, seed=0, width=8, head_scale=0.25)
        dnn = getattr(model.resnet, kind)(model.ConfigChannels(cfg, sd), anchors, C)
    dnn.load_state_dict(sd, strict=False)
    return model.Inference(cfg, dnn, anchors).to(dev()).train(), anchors


def batches(S, B, C=20, onehot=False):
    """Three batches with different maximum box counts (5, 8, 3 rows): the plan pads all of them to one static label buffer."""
    out = []
    for i, nmax in enumerate((5, 8, 3)):
        d = {k: v.to(dev()) for k, v in synth.labels(B, S, C, nmax=nmax, seed=20 + i, onehot=onehot).items()}
        d['tensor'] = synth.images(B, S, seed=30 + i).to(dev())
        out.append(d)
    return out


def run_steps(kind, data, steps, plan, graph, onehot=False, lr=0.0):
    import train as y2train
    import utils
    y2train.PLAN, y2train.GRAPH = plan, graph
    try:
        inf, anchors = build(kind)
        opt = utils.optim.SGD(inf.parameters(), lr, momentum=0.9)
        losses, grads = [], None
        for i in range(steps):
            r = y2train.iterate(inf, opt, data[i % len(data)], oloss.HPARAM, 0.6, anchors)
            losses.append([float(r['loss'][k].detach()) for k in r['loss']] + [float(r['loss_total'].detach())])
            if i == steps - 1:
                grads = {k: p.grad.detach().clone() for k, p in inf.dnn.named_parameters()}
                pred = {k: v.detach().clone() for k, v in r['pred'].items()}
                pos = r['debug']['positive'].clone()
        torch.cuda.synchronize()
        runner = inf.__dict__.get('_y2_step_runner')
        return dict(losses=losses, grads=grads, pred=pred, pos=pos, params={k: v.detach().clone() for k, v in inf.dnn.state_dict().items()}, runner=runner)
    finally:
        y2train.PLAN, y2train.GRAPH = True, True


def assert_same_run(a, b, what, tol=STEP_TOL):
    np.testing.assert_allclose(np.array(a['losses']), np.array(b['losses']), rtol=2e-5, err_msg=what)
    for k in a['params']:
        if a['params'][k].dtype.is_floating_point:
            e = rel(b['params'][k], a['params'][k])
            assert e <= tol, (what, k, e)
        else:
            assert torch.equal(a['params'][k], b['params'][k]), (what, k)          # num_batches_tracked: incremented by every replay
    for k in a['grads']:
        assert rel(b['grads'][k], a['grads'][k]) <= 50 * tol, (what, 'grad', k, rel(b['grads'][k], a['grads'][k]))
    for k in a['pred']:
        assert a['pred'][k].shape == b['pred'][k].shape and rel(b['pred'][k], a['pred'][k]) <= 5 * tol, (what, 'pred', k, rel(b['pred'][k], a['pred'][k]))
    assert torch.equal(a['pos'], b['pos']), what


@pytest.mark.parametrize('onehot', [False, True])
def test_darknet_step_plan_equals_autograd_eager_and_replayed(onehot):
    S, B, steps = 96, 3, 8          # (the last step runs another batch than the first replay did: a replay that kept anything of its first launch would show)
    data = batches(S, B, onehot=onehot)
    run_steps('darknet', data, 4, plan=False, graph=False, onehot=onehot)          # the per-layer algorithm table is measured once: every compared run sees the same one
    ref = run_steps('darknet', data, steps, plan=False, graph=False, onehot=onehot)
    assert ref['runner'] is None
    eager = run_steps('darknet', data, steps, plan=True, graph=False, onehot=onehot)
    assert eager['runner'] is not None and eager['runner'].captures == 0 and not eager['runner'].broken
    assert_same_run(ref, eager, 'plan, eager launches')
    graphed = run_steps('darknet', data, steps, plan=True, graph=True, onehot=onehot)
    r = graphed['runner']
    assert r.captures == 1 and not r.broken and len(r.plans) == 1          # one shape, box counts 5 / 8 / 3 all padded to 16 rows
    plan = next(iter(r.plans.values()))
    assert [op[0] for op in plan.ops] == ['graph'] and plan.calls == steps         # single GPU: the whole step is ONE graph
    assert_same_run(ref, graphed, 'plan, hipGraph replays')


@pytest.mark.parametrize('kind', ['tiny', 'resnet18', 'resnet50'])
def test_other_plugins_step_plan_equals_autograd(kind):
    S, B, steps = 96, 3, 8
    data = batches(S, B)
    run_steps(kind, data, 4, plan=False, graph=False)
    ref = run_steps(kind, data, steps, plan=False, graph=False)
    graphed = run_steps(kind, data, steps, plan=True, graph=True)
    assert graphed['runner'].captures == 1 and not graphed['runner'].broken
    # ~50 batch-statistics BatchNorm layers on 27 samples per channel amplify the atomics' order noise in the ResNets' gradients
    assert_same_run(ref, graphed, kind, tol=STEP_TOL if kind == 'tiny' else 1e-4)


def test_step_plans_per_input_size_and_box_count_share_one_pool():
    """Multi-scale schedule (utils/data.py:135-141) + a batch whose box count outgrows the padded label buffer: one plan per size (a bigger
    box count supersedes it), warm-up counted per size, every plan replaying correctly after the others ran (they share one memory pool)."""
    import train as y2train
    import utils
    inf, anchors = build('darknet')
    opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
    data = {}
    for S in (96, 128):
        for nmax in (6, 24):
            d = {k: v.to(dev()) for k, v in synth.labels(2, S, 20, nmax=nmax, seed=S + nmax).items()}
            d['tensor'] = synth.images(2, S, seed=S).to(dev())
            data[S, nmax] = d
    order = [(96, 6)] * 5 + [(128, 6)] * 5 + [(96, 24)] * 2 + [(128, 24)] * 2 + [(96, 6), (128, 6), (96, 24), (128, 24)] * 2
    first = {}
    for i, key in enumerate(order):
        r = y2train.iterate(inf, opt, data[key], oloss.HPARAM, 0.6, anchors)
        lt = float(r['loss_total'])
        assert np.isfinite(lt), (i, key)
        first.setdefault(key, lt)
    runner = inf.__dict__['_y2_step_runner']
    # ONE plan per size: the batches with 24 box rows superseded the 16-row plans (captured at their FIRST call: the size was already measured),
    # and the 6-row batches then run in the 32-row plans
    assert len(runner.plans) == 2 and runner.captures == 4 and not runner.broken
    assert all(p.ops is not None for p in runner.plans.values())
    assert sorted(p.static['npad'] for p in runner.plans.values()) == [32, 32]
    # an autograd-path witness on the final weights agrees with the next replay of every plan
    for key in data:
        y2train.PLAN = False
        try:
            wit, _ = build('darknet', sd={k: v.clone() for k, v in inf.dnn.state_dict().items()})
            wopt = utils.optim.SGD(wit.parameters(), 0.0)
            w = y2train.iterate(wit, wopt, data[key], oloss.HPARAM, 0.6, anchors)
        finally:
            y2train.PLAN = True
        zopt = utils.optim.SGD(inf.parameters(), 0.0)
        r = y2train.iterate(inf, zopt, data[key], oloss.HPARAM, 0.6, anchors)
        np.testing.assert_allclose(float(r['loss_total']), float(w['loss_total']), rtol=2e-5)
        for (k, a), (_, b) in zip(inf.dnn.named_parameters(), wit.dnn.named_parameters()):
            assert rel(a.grad, b.grad) <= 1e-3, (key, k, rel(a.grad, b.grad))


def test_plans_of_all_sizes_live_in_one_activation_arena_sized_for_the_largest():
    """configs[3] (utils/data.py:135-141 changes the size every `maintain` batches; config.ini:39 lists the sizes up front): train.reserve sizes the arena
    for the largest size - nothing executes: weights, BatchNorm statistics and step counters are untouched - and afterwards neither the eager warm-up passes nor
    the captures of ANY size grow the process's reserved memory by more than the static inputs / gradients a plan owns; results equal the arena-less path."""
    import train as y2train
    import utils
    sizes = (96, 160, 224, 288)
    data = {}
    for S in sizes:
        d = {k: v.to(dev()) for k, v in synth.labels(4, S, 20, nmax=6, seed=S).items()}
        d['tensor'] = synth.images(4, S, seed=S).to(dev())
        data[S] = d
    runs = {}
    for arena in (True, False):
        y2train.ARENA = arena
        try:
            inf, anchors = build('darknet')
            opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
            before = {k: v.clone() for k, v in inf.dnn.state_dict().items()}
            if arena:
                got = y2train.reserve(inf, data[max(sizes)], oloss.HPARAM, 0.6, anchors)
                assert got is not None
                torch.cuda.synchronize()
                for k, v in inf.dnn.state_dict().items():
                    assert torch.equal(v, before[k]), k                                   # a reservation executes nothing
            torch.cuda.synchronize()
            base = torch.cuda.memory_reserved()
            losses = []
            for S in sizes:                      # ascending: every size is larger than all before it - the order that used to grow the pool every time
                for _ in range(5):
                    losses.append(float(y2train.iterate(inf, opt, data[S], oloss.HPARAM, 0.6, anchors)['loss_total']))
            for S in sizes:
                losses.append(float(y2train.iterate(inf, opt, data[S], oloss.HPARAM, 0.6, anchors)['loss_total']))
            torch.cuda.synchronize()
            runner = inf.__dict__['_y2_step_runner']
            assert runner.captures == len(sizes) and not runner.broken and not runner.eager_only
            runs[arena] = (losses, torch.cuda.memory_reserved() - base, {k: v.clone() for k, v in inf.dnn.state_dict().items()})
            if arena:
                assert runner.arena is not None
                # what may still be allocated per plan: its static image batch, label rows, result views and the gradient tensors (~2 x the model); never activations
                own = sum(4 * 3 * S * S * 4 for S in sizes) + len(sizes) * 3 * sum(p.numel() * 4 for p in inf.parameters())
                assert runs[arena][1] <= own + (64 << 20), (runs[arena][1], own)
            del inf, opt, runner
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        finally:
            y2train.ARENA = True
    np.testing.assert_allclose(np.array(runs[True][0]), np.array(runs[False][0]), rtol=3e-4)
    for k, v in runs[True][2].items():
        if v.dtype.is_floating_point:
            assert rel(v, runs[False][2][k]) <= 2e-3, (k, rel(v, runs[False][2][k]))


def test_eval_and_detect_see_the_weights_a_replayed_step_wrote():
    """The replayed graph updates parameters (through the eager optimizer) and BatchNorm buffers (inside the graph, raw pointers):
    the eval-mode caches must follow (version counters advanced per replay)."""
    import model
    import train as y2train
    import utils
    inf, anchors = build('darknet')
    opt = utils.optim.SGD(inf.parameters(), 1e-2, momentum=0.9)
    data = batches(96, 3)
    x = data[0]['tensor']
    for i in range(6):
        y2train.iterate(inf, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
        inf.eval()
        with torch.no_grad():
            got = model._inference(inf, x)['feature'].clone()
        wit, _ = build('darknet', sd={k: v.clone() for k, v in inf.dnn.state_dict().items()})
        wit.eval()
        with torch.no_grad():
            want = model._inference(wit, x)['feature']
        assert torch.equal(got, want), i
        inf.train()


# ------------------------------------------------------------------------------------------------ data parallel, two ranks
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_rank(rank, world, port, tmp, modes, kind='darknet'):
    import sys
    from conftest import APP, ROOT
    for p in (ROOT, APP):
        if p not in sys.path:
            sys.path.insert(0, p)
    rccl = torch.cuda.device_count() >= world and os.environ.get('Y2_TEST_DP_BACKEND', 'auto') != 'gloo'
    if rccl:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        os.environ.pop('Y2_DIST_BACKEND', None)
    else:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', Y2_DIST_BACKEND='gloo')
    import train as y2train
    import utils
    assert y2train.init_distributed() == world
    y2train.PLAN, y2train.GRAPH = modes[rank]
    inf, anchors = build(kind)
    m = y2train.ensure_model(inf)
    assert isinstance(m, y2train.DataParallelRCCL)
    m.bucket_bytes = 4096
    m._build_buckets()                       # many small buckets: several segment cuts inside backward
    opt = utils.optim.SGD(m.parameters(), 0.0, momentum=0.9)          # learning rate 0: every step is the same function of (weights, shard)
    S, B = 96, 4
    per = B // world
    data = []
    for i, nmax in enumerate((5, 8, 3)):
        d = {k: v[rank * per:(rank + 1) * per].cuda() for k, v in synth.labels(B, S, 20, nmax=nmax, seed=20 + i).items()}
        d['tensor'] = synth.images(B, S, seed=30 + i)[rank * per:(rank + 1) * per].cuda()
        data.append(d)
    losses = []
    for i in range(8):
        r = y2train.iterate(m, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
        losses.append([float(r['loss'][k].detach()) for k in r['loss']])
    torch.cuda.synchronize()
    runner = inf.__dict__.get('_y2_step_runner')
    ops = None
    if runner is not None and runner.plans:
        plan = next(iter(runner.plans.values()))
        ops = None if plan.ops is None else [op[0] for op in plan.ops]
    torch.save({'backend': dist.get_backend(), 'device': torch.cuda.current_device(), 'losses': losses, 'params': {k: v.cpu() for k, v in inf.dnn.state_dict().items()}, 'grads': {k: p.grad.cpu() for k, p in inf.dnn.named_parameters()},
                'ops': ops, 'captures': None if runner is None else runner.captures}, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_dp_world2_resnet_graph_segments_equal_the_hook_path(tmp_path):
    """The same protocol for the ResNet plugins (BASELINE configs[4] runs them data parallel): their backward finishes weights in another order
    (`backward_param_order`), writes 1x1 gradients straight into the bucket slices and copies the others there with a kernel."""
    import torch.multiprocessing as mp
    world = 2
    out = {}
    for name, modes in (('hooks', [(False, False)] * 2), ('graphs', [(True, True)] * 2)):
        d = tmp_path / name
        d.mkdir()
        mp.spawn(_dp_rank, args=(world, _free_port(), str(d), modes, 'resnet18'), nprocs=world, join=True)
        out[name] = [torch.load(str(d / ('rank%d.pt' % r)), weights_only=False) for r in range(world)]
    g0 = out['graphs'][0]
    assert g0['captures'] == 1 and g0['ops'].count('graph') >= 3 and 'buckets' in g0['ops'], g0['ops']
    for r in range(world):
        np.testing.assert_allclose(np.array(out['graphs'][r]['losses']), np.array(out['hooks'][r]['losses']), rtol=5e-5)
    for k, v in out['graphs'][0]['grads'].items():
        assert torch.equal(v, out['graphs'][1]['grads'][k]), k
        assert rel(v, out['hooks'][0]['grads'][k]) <= 5e-3, (k, rel(v, out['hooks'][0]['grads'][k]))


@pytest.mark.timeout(900)
def test_dp_world2_graph_segments_interoperate_with_the_hook_path(tmp_path):
    """Two ranks x 8 steps: (a) both on the autograd / hook path, (b) both replaying graph segments, (c) rank 0 on graphs and rank 1 on
    hooks.  Same collectives in the same order in all three: the runs must agree (and not hang)."""
    import torch.multiprocessing as mp
    world = 2
    out = {}
    for name, modes in (('hooks', [(False, False)] * 2), ('graphs', [(True, True)] * 2), ('mixed', [(True, True), (False, False)])):
        d = tmp_path / name
        d.mkdir()
        mp.spawn(_dp_rank, args=(world, _free_port(), str(d), modes), nprocs=world, join=True)
        out[name] = [torch.load(str(d / ('rank%d.pt' % r)), weights_only=False) for r in range(world)]
    g0 = out['graphs'][0]
    assert g0['captures'] == 1 and g0['ops'].count('graph') >= 3 and 'npos' in g0['ops'] and 'buckets' in g0['ops'], g0['ops']
    assert out['mixed'][0]['captures'] == 1 and out['mixed'][1]['captures'] is None
    for name in ('graphs', 'mixed'):
        for r in range(world):
            np.testing.assert_allclose(np.array(out[name][r]['losses']), np.array(out['hooks'][r]['losses']), rtol=2e-5, err_msg='%s rank %d' % (name, r))
            for k, v in out['hooks'][r]['params'].items():
                if v.dtype.is_floating_point and 'running' not in k:
                    assert rel(out[name][r]['params'][k], v) <= STEP_TOL, (name, r, k, rel(out[name][r]['params'][k], v))
        # replicas stay identical: both ranks hold the same averaged gradients and parameters
        for k, v in out[name][0]['grads'].items():
            assert torch.equal(v, out[name][1]['grads'][k]), (name, k)
            assert rel(v, out['hooks'][0]['grads'][k]) <= 50 * STEP_TOL, (name, 'grad', k)


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='real RCCL between two ranks needs two visible GPUs (the 1-GPU lease runs the gloo / shared-GPU variants above)')
def test_real_rccl_world2_captured_steps_equal_the_host_staged_run(tmp_path):
    """train.ensure_model across GPUs (/root/reference/train.py:65-71, loader scaling :296-309) on the transport it ships with: two ranks, one GPU each,
    backend nccl (= RCCL), eight steps of train.iterate - eager plans, the capture, replays of hipGraph segments with RCCL bucket all-reduces and the
    positive-count all-reduce BETWEEN them on the same stream - against the same two ranks sharing cuda:0 over gloo with host-staged buffers (the variant
    every 1-GPU run exercises) and against the hook path on RCCL.  Same collectives in the same order: same losses, same averaged gradients, no hang."""
    import torch.multiprocessing as mp
    world = 2
    out = {}
    for name, modes, backend in (('rccl_graphs', [(True, True)] * 2, 'auto'), ('rccl_hooks', [(False, False)] * 2, 'auto'), ('gloo_graphs', [(True, True)] * 2, 'gloo')):
        d = tmp_path / name
        d.mkdir()
        os.environ['Y2_TEST_DP_BACKEND'] = backend
        try:
            mp.spawn(_dp_rank, args=(world, _free_port(), str(d), modes), nprocs=world, join=True)
        finally:
            os.environ.pop('Y2_TEST_DP_BACKEND', None)
        out[name] = [torch.load(str(d / ('rank%d.pt' % r)), weights_only=False) for r in range(world)]
    g = out['rccl_graphs']
    assert [r['backend'] for r in g] == ['nccl', 'nccl'] and sorted(r['device'] for r in g) == [0, 1]
    assert [r['backend'] for r in out['gloo_graphs']] == ['gloo', 'gloo']
    assert g[0]['captures'] == 1 and g[0]['ops'].count('graph') >= 3 and 'npos' in g[0]['ops'] and 'buckets' in g[0]['ops'], g[0]['ops']
    for other in ('rccl_hooks', 'gloo_graphs'):
        for r in range(world):
            np.testing.assert_allclose(np.array(g[r]['losses']), np.array(out[other][r]['losses']), rtol=2e-5, err_msg='%s rank %d' % (other, r))
        for k, v in g[0]['grads'].items():
            assert torch.equal(v, g[1]['grads'][k]), k                          # replicas hold the same averaged gradient, bit for bit
            assert rel(v, out[other][0]['grads'][k]) <= 50 * STEP_TOL, (other, k)


def test_plans_are_recaptured_after_the_model_moved_to_the_cpu_and_back():
    """train.py:423-432: Train.eval moves the inference module to the CPU and back between training steps.  The Parameters are the same
    objects afterwards but live in new memory: a captured graph would read the old addresses.  The runner must notice and re-capture."""
    import train as y2train
    import utils
    inf, anchors = build('darknet')
    opt = utils.optim.SGD(inf.parameters(), 0.0, momentum=0.9)
    data = batches(96, 3)
    for i in range(5):
        r = y2train.iterate(inf, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
    runner = inf.__dict__['_y2_step_runner']
    assert runner.captures == 1
    before = [float(r['loss'][k].detach()) for k in r['loss']]
    g_before = {k: p.grad.detach().clone() for k, p in inf.dnn.named_parameters()}
    old = [t.data for t in list(inf.parameters()) + list(inf.buffers())]      # held: the round trip cannot land on the same addresses
    inf.cpu()
    inf.cuda()
    for t in old:
        if t.dtype.is_floating_point:
            t.fill_(float('nan'))                                          # whoever still reads the old memory computes NaN
    opt = utils.optim.SGD(inf.parameters(), 0.0, momentum=0.9)
    for i in range(5):                                                    # same batch as the step whose results were kept (i = 4 -> data[1])
        r = y2train.iterate(inf, opt, data[1], oloss.HPARAM, 0.6, anchors)
    assert runner.captures == 2 and not runner.broken
    after = [float(r['loss'][k].detach()) for k in r['loss']]
    np.testing.assert_allclose(after, before, rtol=2e-5)
    for k, p in inf.dnn.named_parameters():
        assert rel(p.grad, g_before[k]) <= 1e-3, k
    del old


@pytest.mark.parametrize('B,HW,cin,cout', [(64, 52, 128, 256), (16, 104, 64, 128), (64, 13, 512, 1024)])
@pytest.mark.parametrize('flags', [1, 3])
def test_winograd_weight_gradient_is_replay_safe(B, HW, cin, cout, flags):
    """A captured library call must re-establish everything it reads: the Winograd weight gradients zero their split accumulators first - with
    hipMemsetAsync until round 4, and captured in front of the accumulating kernels the memset node did its job at the graph's FIRST launch only on this
    runtime: the first replay of a training step was right, every later one accumulated onto whatever the scratch held.  Replays on changing inputs with the scratch poisoned in
    between must equal eager calls."""
    import _hip
    L = _hip.lib()
    d = dev()
    x = torch.randn(B, HW, HW, cin, device=d)
    dz = torch.randn(B, HW, HW, cout, device=d) * 1e-3
    ws = torch.empty(L.y2_wino_wgrad_workspace_bytes(B, HW, HW, cin, cout) // 4 + 4, device=d)
    out, ref = torch.empty(cout, cin, 3, 3, device=d), torch.empty(cout, cin, 3, 3, device=d)

    def call(dst):
        _hip.check(L.y2_wino_wgrad_ex(_hip.ptr(x), _hip.ptr(dz), _hip.ptr(dst), B, HW, HW, cin, cin, cout, cout, None, _hip.ptr(ws), ws.numel() * 4, flags, _hip.stream()), 'wgrad')
    call(ref)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g.capture_begin()
        call(out)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    for rep in range(3):
        x.normal_()
        dz.normal_().mul_(1e-3)
        ws.fill_(float('nan') if rep == 1 else 1e30)
        g.replay()
        torch.cuda.synchronize()
        got = out.clone()
        ws.zero_()
        call(ref)
        torch.cuda.synchronize()
        assert rel(got, ref) <= 1e-4, (rep, rel(got, ref))         # (split partial sums are added atomically: not bit-equal)


def test_full_width_replays_equal_the_autograd_step():
    """The narrow networks above never split a Winograd weight-gradient reduction; the full-width network does (that is where the memset bug hid).
    Full-width Darknet-19, 416x416, batch 4: the 6th step of train.iterate (its 2nd and 3rd REPLAY, learning rate 0) against the autograd path on
    the same batches."""
    import model
    import model.yolo2
    import train as y2train
    import utils
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    sd = odark.init_state_dict(5, 20, seed=0, head_scale=1 / 40.0)
    data = []
    for i in range(2):
        d = {k: v.to(dev()) for k, v in synth.labels(4, 416, 20, nmax=6, seed=40 + i).items()}
        d['tensor'] = synth.images(4, 416, seed=50 + i).to(dev())
        data.append(d)
    runs = {}
    for plan in (False, True):
        y2train.PLAN = plan
        try:
            dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, 20)
            dnn.load_state_dict(sd, strict=False)
            inf = model.Inference(cfg, dnn, anchors).to(dev()).train()
            opt = utils.optim.SGD(inf.parameters(), 0.0)
            rows = []
            for i in range(7):
                r = y2train.iterate(inf, opt, data[i % 2], oloss.HPARAM, 0.6, anchors)
                rows.append(([float(r['loss'][k].detach()) for k in r['loss']], {k: p.grad.detach().clone() for k, p in dnn.named_parameters()} if i >= 5 else None))
            torch.cuda.synchronize()
            runs[plan] = rows
            if plan:
                assert inf.__dict__['_y2_step_runner'].captures == 1
        finally:
            y2train.PLAN = True
    # What "equal" means at full width (README "Tolerances"): the weight gradients in front of a batch-statistics BatchNorm are cancellation
    # residues - the oracle's own fp32 run is 0.5-0.9 % (rms) away from its fp64 run there, and two fp32 runs that add in different orders
    # (completion-order atomics: the autograd path and a replay, or two replays) differ by about that floor.  So both paths are held to the
    # ORACLE as a coarse anchor (10 x the floor; the measured ratios are printed) and to EACH OTHER with rms error <= max(1e-3, 2.5 x floor);
    # a wrong replay (stale scratch, a missing zero fill) is off by O(1).
    from oracle import head as ohead
    torch.set_num_threads(64)

    def rms_rel(got, ref):
        ref = ref.double().cpu()
        return ((got.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)).item()
    ref = {}
    for b in range(2):
        x = data[b]['tensor'].cpu()
        nd = synth.norm_data({k: v.cpu() for k, v in data[b].items() if k != 'tensor'}, 416, 416, 13, 13)
        for name, dt in (('fp64', torch.float64), ('fp32', torch.float32)):
            sdx = {k: (v.detach().clone().to(dt).requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}      # (clone: `.to(float32)` of an fp32 tensor is the tensor itself)
            f = odark.forward(x.to(dt), sdx, training=True)
            lo, _ = oloss.loss(anchors.to(dt), {k: (v.to(dt) if v.is_floating_point() else v) for k, v in nd.items()}, ohead.decode(f, anchors.to(dt)), 0.6)
            oloss.total(lo).backward()
            ref[(b, name)] = {k: v.grad for k, v in sdx.items() if getattr(v, 'grad', None) is not None}
    worst = [0.0, 0.0, 0.0]
    for i, ((la, ga), (lb, gb)) in enumerate(zip(runs[False], runs[True])):
        np.testing.assert_allclose(lb, la, rtol=2e-5, err_msg='step %d' % i)
        if ga is not None:
            g64, g32 = ref[(i % 2, 'fp64')], ref[(i % 2, 'fp32')]
            assert set(ga) == set(g64)
            for k in ga:
                floor = rms_rel(g32[k], g64[k])
                ea, eb, ab = rms_rel(ga[k], g64[k]), rms_rel(gb[k], g64[k]), rms_rel(gb[k], ga[k])
                worst = [max(worst[0], ea / max(4e-5, floor)), max(worst[1], eb / max(4e-5, floor)), max(worst[2], ab / max(4e-4, floor))]
                # (both paths against the oracle are PRINTED, not asserted, here: the floor is one fp32 run of the oracle - itself a draw from that
                # noise - and on these batch-4 draws single BatchNorm parameters of either path land at 2.6-3.5 x it from run to run; the
                # calibrated 2.5 x rule is asserted at batch 4 / seed 3 on the autograd path (tests/test_gpu_fullsize.py) and at batch 64 on
                # the REPLAYED path (tests/test_gpu_b64.py))
                assert ea <= max(1e-4, 10 * floor) and eb <= max(1e-4, 10 * floor), ('far from the oracle', i, k, ea, eb, floor)
                assert ab <= max(1e-3, 2.5 * floor), ('replay vs autograd', i, k, ab, floor)
    print('full-width batch-4 steps: autograd / replay error over the fp32 floor %.2f / %.2f, replay vs autograd %.2f' % tuple(worst))


def test_a_parameter_frozen_after_the_first_steps_drops_the_plans():
    """A plan snapshots the parameter list at creation and writes every gradient: freezing a layer later (requires_grad = False, a
    fine-tuning schedule) must take the step off the captured plans - the frozen parameter receives no gradient any more and does not move."""
    import train as y2train
    import utils
    inf, anchors = build('darknet')
    opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.0)
    data = batches(96, 2)
    for i in range(6):
        y2train.iterate(inf, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
    runner = inf.__dict__['_y2_step_runner']
    assert runner.captures == 1 and len(runner.plans) == 1
    frozen = inf.dnn.layers1[2].conv.weight
    frozen.requires_grad_(False)
    frozen.grad = None
    before = frozen.detach().clone()
    other = inf.dnn.layers1[4].conv.weight
    moved = other.detach().clone()
    r = y2train.iterate(inf, opt, data[0], oloss.HPARAM, 0.6, anchors)
    torch.cuda.synchronize()
    assert np.isfinite(float(r['loss_total']))
    assert len(runner.plans) == 0                      # no plan holds the stale parameter list
    assert frozen.grad is None and torch.equal(frozen.detach(), before)
    assert other.grad is not None and not torch.equal(other.detach(), moved)


def test_capture_falls_back_to_all_operand_forms_when_the_pruned_set_is_short():
    """A captured step prepares only the operand forms its last eager pass read (train_graph._train_operands(only=...)).  If the algorithm
    table asks for another form under capture, the capture is repeated with all of them - never a wrong or missing operand."""
    import train as y2train
    import utils
    from model import train_graph
    sd = odark.init_state_dict(5, 20, seed=0, channels={k: max(v, 32) for k, v in dict(NARROW, **{'layers1.5': 32}).items()}, head_scale=1 / 8.0)     # >= 32 channels: Winograd-eligible layers
    data = batches(96, 2)
    outs = {}
    for mode in ('pruned', 'short', 'off'):
        inf, anchors = build('darknet', sd={k: v.clone() for k, v in sd.items()})
        opt = utils.optim.SGD(inf.parameters(), 0.0)
        train_graph.PRUNE_OPERANDS = mode != 'off'
        try:
            for i in range(3):
                y2train.iterate(inf, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
            runner = inf.__dict__['_y2_step_runner']
            plan = next(iter(runner.plans.values()))
            assert plan.ops is None and plan.used_last
            if mode == 'short':
                keep = sorted(plan.used_last, key=lambda e: (str(id(e[0])), e[1]))[:1]
                plan.used_last = set(keep)                       # the capture will ask for operands outside this set
            for i in range(3, 6):
                r = y2train.iterate(inf, opt, data[i % 3], oloss.HPARAM, 0.6, anchors)
            torch.cuda.synchronize()
            assert runner.captures == 1 and not runner.broken and plan.capture_error is None
            assert (plan.only is not None) == (mode == 'pruned'), (mode, plan.only)
            outs[mode] = ([float(r['loss'][k].detach()) for k in r['loss']], {k: p.grad.detach().clone() for k, p in inf.dnn.named_parameters()})
        finally:
            train_graph.PRUNE_OPERANDS = True
    for mode in ('short', 'off'):
        np.testing.assert_allclose(outs[mode][0], outs['pruned'][0], rtol=2e-5)
        for k in outs['pruned'][1]:
            assert rel(outs[mode][1][k], outs['pruned'][1][k]) <= 1e-3, (mode, k)
