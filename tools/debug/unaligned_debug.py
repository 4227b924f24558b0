"""Per-parameter gradient errors of the unaligned-width training step variants (round-3 debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import model, _hip
from model import train_graph
from oracle import darknet as odark, loss as oloss, synth
from oracle.make_golden import NARROW
import test_gpu_round3 as T

EARLY = {'layers1.0': 6, 'layers1.2': 10, 'layers1.5': 6, 'layers1.16': 30}
for name, widths, bn, streams, tune in [('early/nobn', EARLY, False, 2, True), ('early/bn', EARLY, True, 2, True)]:
    train_graph.BWD_STREAMS = streams
    _hip.AUTOTUNE = tune
    w = dict(NARROW); w['layers1.5'] = 8; w.update(widths)
    sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0, bn=bn)
    inf, anchors = T.build(sd, bn=bn)
    inf.train()
    S, B = 96, 3
    x = synth.images(B, S, seed=1)
    data = synth.norm_data(synth.labels(B, S, 20, seed=2), S, S, 3, 3)
    pred = model._inference(inf, x.to('cuda:0'))
    loss, _ = model.loss(anchors, data, pred, 0.6)
    model.weighted_total(loss, oloss.HPARAM).backward()
    torch.cuda.synchronize()
    sd64, lo, stats, f = T.oracle_step(sd, x, data, anchors, True)
    ours = dict(inf.dnn.named_parameters())
    sd32 = T.oracle_step_fp32(sd, x, data, anchors, True)
    errs = [(k, T.rel(ours[k].grad, v.grad)) for k, v in sd64.items() if v.requires_grad and k.endswith('conv.weight')]
    print('   fp32 torch-CPU oracle vs fp64:', ' '.join('%s=%.0e' % (k.replace('.conv.weight', '').replace('layers', 'L'), T.rel(sd32[k].grad, v.grad)) for k, v in sd64.items() if v.requires_grad and k.endswith('conv.weight')))
    print(name, 'feature err %.2e' % T.rel(pred['feature'], f.detach()), '| weight-grad errors in layer order:', ' '.join('%s=%.0e' % (k.replace('.conv.weight', '').replace('layers', 'L'), e) for k, e in errs))
