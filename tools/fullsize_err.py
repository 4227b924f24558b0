#!/usr/bin/env python
"""Worst-case feature error (max |err| / rms against the fp64 oracle, 8 sampled images of the batch-32 416x416 run) of the whole-model
forward under the autotuned plan and under each forced algorithm plan - the quantity tests/test_gpu_fullsize.py bounds."""
import configparser
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402
import model  # noqa: E402
import model.yolo2  # noqa: E402
from oracle import darknet as odark, synth  # noqa: E402

SAMPLED = (0, 3, 7, 13, 18, 22, 27, 31)
dev = torch.device('cuda:0')
cfg = configparser.ConfigParser()
cfg.read_dict({'batch_norm': {'enable': '1'}})
anchors = torch.from_numpy(synth.ANCHORS_VOC)
sd = odark.init_state_dict(5, 20, seed=0, head_scale=1 / 40.0)
dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, 20)
dnn.load_state_dict(sd, strict=False)
inf = model.Inference(cfg, dnn, anchors).to(dev).eval()
x = synth.images(32, 416, seed=1)
torch.set_num_threads(min(64, os.cpu_count() or 8))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
with torch.no_grad():
    truth = odark.forward(x[list(SAMPLED)].double(), sd64)
rms = truth.pow(2).mean((1, 2, 3)).sqrt()
for mode in (None, None, None, 'direct', 'winograd', 'fused', 'implicit'):
    _hip.FORCE_ALGO = mode
    if mode is None:
        _hip._TUNE.clear()
    inf.dnn._plan_cache = None
    with torch.no_grad():
        f = inf.dnn.forward_nhwc(x.to(dev))[list(SAMPLED)].permute(0, 3, 1, 2).double().cpu()
    plan = inf.dnn._plan_cache[1]
    algos = [plan['arr'][i].algo for i in range(plan['n'])]
    err = ((f - truth).abs().amax((1, 2, 3)) / rms)
    print('%-9s worst %.3e  per image %s  algos %s' % (mode or 'autotune', err.max().item(), ' '.join('%.2e' % e for e in err.tolist()), ''.join(str(a) for a in algos)), flush=True)
