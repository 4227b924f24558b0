#!/usr/bin/env python
"""Measure the per-layer algorithm choices for the BASELINE shapes on this MI355X and write them as the committed default table
(yolo2-pytorch_amd/tune/default_gfx950.json, keyed by the hash of the kernel sources): detect at batch 1 / 8 / 32 and training at batch 64,
Darknet-19 with 20 and 80 classes at every size of the multi-scale schedule (config.ini:39-40), plus ResNet-50 at 608x608 / 80 classes.
A fresh process then starts warm: `first_visit_ms` of the multi-scale leg drops from seconds to the cost of one untimed step."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='320,352,384,416,448,480,512,544,576,608')
    ap.add_argument('--classes', default='20,80')
    ap.add_argument('--no-resnet', action='store_true')
    args = ap.parse_args()
    os.environ['Y2_TUNE_DEFAULTS'] = '0'          # measure everything afresh
    import torch

    import _hip
    import bench_data
    import detect
    import train as y2train
    import utils
    dev = torch.device('cuda', 0)
    y2train.GRAPH = False
    t0 = time.time()
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
    sizes = [int(v) for v in args.sizes.split(',')]

    def detect_shapes(inf, anchors, shapes):
        inf.eval()
        for B, S in shapes:
            x = bench_data.images(B, S, seed=1).to(dev)
            with torch.no_grad():
                for _ in range(3):
                    detect.detect_batch(inf.dnn.forward_nhwc(x), anchors, **kw)
            torch.cuda.synchronize()

    def train_shapes(inf, anchors, C, shapes):
        inf.train()
        opt = utils.optim.SGD(inf.parameters(), 0.0)
        for B, S in shapes:
            d = {k: v.to(dev) for k, v in bench_data.labels(B, S, C, seed=2).items()}
            d['tensor'] = bench_data.images(B, S, seed=11).to(dev)
            # the choices of a shape settle over several steps (the weight gradient is measured in the first backward with the forward's transformed
            # input at hand; where it then runs a form that does not read it the next forward is a NEW problem - no transformed input to keep -
            # and so is that layer's next weight gradient): iterate until two consecutive steps measured nothing
            quiet, steps = 0, 0
            while quiet < 2 and steps < 10:
                before = len(_hip.TUNE_MISSES)
                y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
                torch.cuda.synchronize()
                steps += 1
                quiet = quiet + 1 if len(_hip.TUNE_MISSES) == before else 0
            print('  train B=%d S=%d C=%d: %d entries after %d steps, %.0f s' % (B, S, C, len(_hip._TUNE), steps, time.time() - t0), flush=True)

    for C in [int(v) for v in args.classes.split(',')]:
        inf, anchors = bench_data.build_model(C, dev, 'darknet')
        detect_shapes(inf, anchors, [(32, S) for S in sizes] + ([(1, 416), (8, 416), (64, 416)] if C == 20 else []))
        print('detect C=%d: %d entries, %.0f s' % (C, len(_hip._TUNE), time.time() - t0), flush=True)
        train_shapes(inf, anchors, C, [(64, S) for S in sizes])
        del inf
        torch.cuda.empty_cache()
    if not args.no_resnet:
        inf, anchors = bench_data.build_model(80, dev, 'resnet50')
        detect_shapes(inf, anchors, [(32, 608)])
        train_shapes(inf, anchors, 80, [(32, 608)])
    n = _hip.save_tune_defaults(note='tools/make_tune_table.py on %s, torch %s' % (torch.cuda.get_device_name(0), torch.__version__))
    print('wrote %s: %d entries for kernels %s in %.0f s' % (_hip.DEFAULTS_PATH, n, _hip.kernel_hash(), time.time() - t0))


if __name__ == '__main__':
    main()
