"""BASELINE configs[2] AT ITS OWN SHAPE on the code path that produces the benchmark number.

The reference's training recipe is `train.py -b 64` (quick_start.sh:71) through `Train.iterate` (train.py:338-362).  bench.py times
`train.iterate` at batch 64, 416x416, VOC-20: after three eager plan steps the step is captured and every later step is a hipGraph REPLAY of
model.train_graph.StepPlan.  Tile shapes, split-K factors and the Winograd forms a layer takes all depend on the batch size, and round 4's
memset bug lived in replay-only, full-width-only behaviour - so this test holds exactly that: full-width Darknet-19, batch 64, the REPLAYED
step, on two alternating batches, against the oracle's fp64 autograd of the same step (oracle/darknet.py, oracle/loss.py), with the oracle's
own fp32 run as the arithmetic floor (the production rule of tests/test_gpu_fullsize.py).  Learning rate 0: every step is the same function
of (weights, batch), so the replays of one batch must agree with each other, too.  Both forms of the captured step are held to it: the
single-process default (forked graph, pruned operand preparation) and the linear graph a rank under the data-parallel wrapper replays."""
import configparser
import os
import time

import numpy as np
import pytest
import torch

from oracle import darknet as odark
from oracle import head as ohead
from oracle import loss as oloss
from oracle import synth

pytestmark = pytest.mark.gpu

B, S, C = 64, 416, 20


def dev():
    return torch.device('cuda', 0)


def rms_rel(got, ref):
    ref = ref.double()
    return ((got.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)).item()


def oracle_step(sd, anchors, x, data, dt):
    """fwd (batch statistics) + region loss + backward of the oracle in dtype `dt`: (loss terms, parameter gradients)."""
    # (detach + clone: `.to(float32)` of an fp32 tensor is the tensor itself - requires_grad_ would mark the caller's state dict)
    sdx = {k: (v.detach().clone().to(dt).requires_grad_('running' not in k) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    f = odark.forward(x.to(dt), sdx, training=True)
    lo, _ = oloss.loss(anchors.to(dt), {k: (v.to(dt) if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(f, anchors.to(dt)), 0.6)
    oloss.total(lo).backward()
    return {k: float(v.detach()) for k, v in lo.items()}, {k: v.grad for k, v in sdx.items() if getattr(v, 'grad', None) is not None}


def replayed_steps(sd, anchors, data, fork, prune):
    """Ten train.iterate steps at learning rate 0 over the two batches: {batch: [(loss terms, gradients, 'capture' | 'replay'), ...]} + the plan."""
    import model
    import model.yolo2
    import train as y2train
    import utils
    from model import train_graph
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    saved = (train_graph.GRAPH_FORK, train_graph.PRUNE_OPERANDS)
    train_graph.GRAPH_FORK, train_graph.PRUNE_OPERANDS = fork, prune
    inf = None
    try:
        dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, C)
        dnn.load_state_dict(sd, strict=False)
        inf = model.Inference(cfg, dnn, anchors).to(dev()).train()
        opt = utils.optim.SGD(inf.parameters(), 0.0)          # the fused optimizer still runs every step (y2_opt_sgd), it just moves nothing
        # ---- 3 eager plan steps, the capture (4th call, replayed right away), then 6 more replays: 7 replays over the two batches
        seen = {0: [], 1: []}
        for i in range(10):
            r = y2train.iterate(inf, opt, data[i % 2], oloss.HPARAM, 0.6, anchors)
            runner = inf.__dict__['_y2_step_runner']
            if runner.last[1] == 'replay' or (i >= 3 and runner.last[1] == 'capture'):
                seen[i % 2].append(({k: float(r['loss'][k].detach()) for k in r['loss']}, {k: p.grad.detach().clone().cpu() for k, p in dnn.named_parameters()}, runner.last[1]))
        torch.cuda.synchronize()
        assert runner.captures == 1 and runner.broken is None and not runner.eager_only, (runner.captures, runner.broken, runner.eager_only)
        assert all(len(v) >= 3 for v in seen.values()), {k: [m for _, _, m in v] for k, v in seen.items()}
        assert sum(1 for v in seen.values() for _, _, m in v if m == 'replay') >= 6
        for p in dnn.parameters():
            assert torch.isfinite(p.grad).all()
        return seen, next(iter(runner.plans.values()))
    finally:
        train_graph.GRAPH_FORK, train_graph.PRUNE_OPERANDS = saved
        del inf
        torch.cuda.empty_cache()


@pytest.mark.timeout(1500)
def test_batch64_replayed_training_step_equals_the_oracle():
    import train as y2train
    assert y2train.PLAN and y2train.GRAPH, 'this test is about the captured path (Y2_TRAIN_PLAN / Y2_TRAIN_GRAPH must be on)'
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    sd = odark.init_state_dict(5, C, seed=0, head_scale=1 / 40.0)
    host, data = [], []
    for i in range(2):
        lab = synth.labels(B, S, C, nmax=8, seed=60 + i)
        x = synth.images(B, S, seed=70 + i)
        host.append((x, synth.norm_data(lab, S, S, S // 32, S // 32)))
        d = {k: v.to(dev()) for k, v in lab.items()}
        d['tensor'] = x.to(dev())
        data.append(d)
    # the two forms a captured step takes: single-process default (weight gradients on a forked branch of the graph, only the GEMM-operand forms
    # the layers read prepared per step) and what a rank under the data-parallel wrapper replays (ONE linear chain; here also with every operand form)
    runs = {}
    for name, fork, prune in (('forked graph, pruned operands', 'auto', True), ('linear graph, all operand forms', False, False)):
        seen, plan = replayed_steps(sd, anchors, data, fork, prune)
        if prune:
            # the captured step derives only the GEMM-operand forms its layers' chosen algorithms read (one of {packed, Winograd} per pass and layer)
            assert plan.only is not None, 'the pruned capture fell back to preparing every operand form'
            per_layer = {}
            for mod, tag in plan.only:
                per_layer.setdefault(mod, set()).add(tag)
            assert len(per_layer) >= 21 and all(          # (every block but the first; the 125-channel head's data gradient runs zero-padded outside the prepared set)
                len(v & {'wp', 'uf'}) == 1 and len(v & {'wd', 'ud', 'u6d'}) == 1 for v in per_layer.values()), sorted(map(sorted, per_layer.values()))
            print('operand forms prepared per step: %s' % {t: sum(1 for _, tag in plan.only if tag == t) for t in ('wp', 'uf', 'wd', 'ud', 'u6d')})
        else:
            assert plan.only is None
        # ---- replays of the same batch agree with each other (what differs: completion-order atomics of split reductions and BatchNorm sums)
        for b, rows in seen.items():
            for lo, gr, _ in rows[1:]:
                for k in lo:
                    np.testing.assert_allclose(lo[k], rows[0][0][k], rtol=2e-5, err_msg='%s: batch %d loss %s between replays' % (name, b, k))
        runs[name] = seen
    # ---- against the oracle
    torch.set_num_threads(max(1, min(96, os.cpu_count() or 1)))
    worst = 0.0
    for b in (0, 1):
        x, nd = host[b]
        t0 = time.time()
        l64, g64 = oracle_step(sd, anchors, x, nd, torch.float64)
        t1 = time.time()
        if b == 0:
            # the fp32 floor is a property of the arithmetic at this shape (the same network, batch size and label statistics): measured on
            # the first batch, applied to both (one oracle pass less: the fp64 passes are 40 s each on the GPU box's host)
            l32, g32 = oracle_step(sd, anchors, x, nd, torch.float32)
            floors = {k: rms_rel(g32[k], g64[k]) for k in g64}
        print('oracle batch-%d step: fp64 %.1f s%s' % (B, t1 - t0, ', fp32 %.1f s' % (time.time() - t1) if b == 0 else ''))
        for name, seen in runs.items():
            lo, gr, mode = seen[b][-1]
            assert mode == 'replay'
            assert set(lo) == set(l64) and len(lo) == 5
            for k in lo:
                np.testing.assert_allclose(lo[k], l64[k], rtol=1e-5, err_msg='%s: batch %d loss %s' % (name, b, k))
            assert set(gr) == set(g64)
            rows = []
            for k in g64:
                floor = floors[k]
                e = rms_rel(gr[k], g64[k])
                rows.append((e / max(1e-4 / 2.5, floor), k, e, floor))
                # the other replays of this batch are the same step: hold them to the same bound
                for _, gother, _ in seen[b][:-1]:
                    eo = rms_rel(gother[k], g64[k])
                    assert eo <= max(1e-4, 2.5 * floor), (name, 'earlier replay', b, k, eo, floor)
            rows.sort(reverse=True)
            print('%s - batch %d of 2, REPLAYED batch-64 step vs fp64 oracle: worst gradient error / fp32 floor = %.2f' % (name, b, rows[0][0]))
            for r in rows[:3]:
                print('    %-28s error %.3e  fp32 floor %.3e  ratio %.2f' % (r[1], r[2], r[3], r[0]))
            worst = max(worst, rows[0][0])
            assert rows[0][0] <= 2.5, (name,) + rows[0]
    print('batch-64 replayed step: worst ratio %.2f (bound 2.5)' % worst)
