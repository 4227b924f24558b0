#!/usr/bin/env python
"""In-step verification of the algorithm table for ONE training shape (default: Darknet-19 VOC-20, batch 64, 416x416).

The per-layer measurements of _hip.autotune_conv / _hip.conv_wgrad time a candidate alone, back to back; in the step a kernel meets another cache state
(a gradient the previous kernel has just left in the Infinity Cache, operands of the next one evicted) and co-running weight gradients, and near-ties can
resolve the wrong way (the 208x208 weight gradient: 6 % ahead alone, 0.11 ms behind in the step).  This tool takes the committed table as the starting
point and does one pass of coordinate descent with the TIMED captured step as the objective: for every table entry the step reads, each alternative is
tried (plans are rebuilt: 3 eager passes + capture + timed replays) and kept if it is faster by more than the noise margin.

    python tools/tune_instep.py [--write] [--steps 30] [--margin 0.04] [--only wgrad|conv|all]

--write: store the improved choices in yolo2-pytorch_amd/tune/default_gfx950.json (same kernel hash), with a note."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _hip  # noqa: E402
import bench_data  # noqa: E402
import train as y2train  # noqa: E402
import utils  # noqa: E402


class Recording(dict):
    """The table, remembering which keys a step looked up."""
    seen = None

    def get(self, key, default=None):
        if self.seen is not None:
            self.seen.append(key)
        return dict.get(self, key, default)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--margin', type=float, default=0.04, help='ms a change must gain to be kept')
    ap.add_argument('--only', default='all', choices=['all', 'wgrad', 'conv'])
    ap.add_argument('--min-hw', type=int, default=0, help='only entries of maps with at least this many pixels per side')
    ap.add_argument('--write', action='store_true')
    ap.add_argument('--detect', action='store_true', help='the batch-32 detect step (BASELINE configs[1]: conv stack + decode + filter + NMS, one captured graph, serial replays) instead of the training step')
    args = ap.parse_args()
    if args.detect and args.batch == 64:
        args.batch = 32
    dev = torch.device('cuda', 0)
    _hip.load_tune_defaults(dev)
    table = Recording(_hip._TUNE)
    _hip._TUNE = table
    inf, anchors = bench_data.build_model(args.classes, dev, 'darknet')
    if args.detect:
        return detect_descent(args, dev, table, inf, anchors)
    inf.train()
    opt = utils.optim.SGD(inf.parameters(), 0.0, momentum=0.9)          # learning rate 0: every trial runs the same arithmetic on the same weights
    data = {k: v.to(dev) for k, v in bench_data.labels(args.batch, args.size, args.classes, seed=2).items()}
    data['tensor'] = bench_data.images(args.batch, args.size, seed=11).to(dev)

    def step():
        return y2train.iterate(inf, opt, data, bench_data.HPARAM, bench_data.THRESHOLD, anchors)

    def measure():
        """ms per replayed step with the current table (plans are rebuilt when the tune epoch moved)."""
        for _ in range(6):
            step()
        runner = inf.__dict__['_y2_step_runner']
        if runner.broken or runner.eager_only or not runner.captures:
            inf.__dict__.pop('_y2_step_runner', None)          # (a choice the capture could not take: start the next trial from a clean runner)
            return float('inf')
        torch.cuda.synchronize()
        best = float('inf')
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.steps)
        return best

    table.seen = []
    step()
    step()
    torch.cuda.synchronize()
    keys = []
    for k in table.seen:
        if k in table and k not in keys:
            keys.append(k)
    table.seen = None
    base = measure()
    print('baseline %.3f ms per step, %d table entries read by the step' % (base, len(keys)), flush=True)
    changed = []
    t0 = time.time()
    for k in keys:
        cur = table[k]
        is_w = k[0] == 'wgrad'
        hw = k[2] if is_w else k[1]
        if (args.only == 'wgrad' and not is_w) or (args.only == 'conv' and is_w) or hw < args.min_hw:
            continue
        if is_w:
            alts = [c for c in ((0, 2, 1) if k[8] else (0, 2)) if c != cur]
        else:
            algo, tile = (tuple(cur) if isinstance(cur, (list, tuple)) else (0, cur))
            ksize, wino_ok, implicit_ok, f43 = k[6], k[18], k[19], ('f43' in k)
            cands = [(0, t) for t in (1, 2, 3, 5)]
            if ksize == 3 and wino_ok:
                cands += [(1, 5), (1, 3), (2, 0), (2, 3)] + ([(3, 0), (3, 3)] if implicit_ok else []) + ([(6, 5), (6, 3)] if f43 else [])
            alts = [c for c in cands if tuple(c) != (algo, tile)]
        for alt in alts:
            table[k] = alt if is_w else list(alt)
            _hip._TUNE_EPOCH[0] += 1
            try:
                t = measure()
            except Exception as e:          # a choice the library refuses for this problem
                t = float('inf')
                print('   %s -> %s: %s' % (k[:8], alt, str(e)[:80]), flush=True)
                inf.__dict__.pop('_y2_step_runner', None)
            if t < base - args.margin:
                print('KEEP %s: %s -> %s  %.3f -> %.3f ms' % (list(k[:10]), cur, alt, base, t), flush=True)
                changed.append((k, cur, alt, base, t))
                base, cur = t, (alt if is_w else list(alt))
            else:
                table[k] = cur
                _hip._TUNE_EPOCH[0] += 1
    final = measure()
    print('after one pass: %.3f ms per step, %d entries changed, %.0f s' % (final, len(changed), time.time() - t0), flush=True)
    if args.write and changed:
        _hip._TUNE = dict(table)
        path = _hip.DEFAULTS_PATH
        old = json.load(open(path))
        n = _hip.save_tune_defaults(note=old.get('note', '') + '; %d entries re-decided in the timed step by tools/tune_instep.py (batch %d, %dx%d)' % (len(changed), args.batch, args.size, args.size))
        print('wrote %s (%d entries)' % (path, n))


def detect_descent(args, dev, table, inf, anchors):
    import detect
    inf.eval()
    dnn = inf.dnn
    x = bench_data.images(args.batch, args.size, seed=1).to(dev)
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)

    def measure():
        dnn._plan_cache = None                        # the plan is rebuilt on the current table
        run = detect.GraphedDetector(dnn, anchors, x, static_input=True, **kw)
        for _ in range(5):
            run.run()
        torch.cuda.synchronize()
        best = float('inf')
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                run.run()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.steps)
        return best
    table.seen = []
    with torch.no_grad():
        detect.detect_batch(dnn.forward_nhwc(x), anchors, **kw)
    torch.cuda.synchronize()
    keys = []
    for k in table.seen:
        if k in table and k not in keys and k[0] != 'wgrad':
            keys.append(k)
    table.seen = None
    base = measure()
    print('baseline %.4f ms per detect step (serial replays), %d table entries read' % (base, len(keys)), flush=True)
    changed = []
    for k in keys:
        cur = table[k]
        algo, tile = (tuple(cur) if isinstance(cur, (list, tuple)) else (0, cur))
        ksize, wino_ok, implicit_ok = k[6], k[18], k[19]
        cands = [(0, t) for t in (1, 2, 3, 5)] + ([(0, t) for t in (11, 12, 13, 15)] if ksize == 1 and not k[8] else [])
        if ksize == 3 and wino_ok:
            cands += [(1, 5), (1, 3), (1, 2), (2, 0), (2, 3)] + ([(3, 0), (3, 3)] if implicit_ok else [])
        for alt in [c for c in cands if tuple(c) != (algo, tile)]:
            table[k] = list(alt)
            try:
                t = measure()
            except Exception as e:
                t = float('inf')
            if t < base - args.margin:
                print('KEEP %s: %s -> %s  %.4f -> %.4f ms' % (list(k[:10]), cur, alt, base, t), flush=True)
                changed.append((k, cur, alt, base, t))
                base, cur = t, list(alt)
            else:
                table[k] = cur
    final = measure()
    print('after one pass: %.4f ms per detect step, %d entries changed' % (final, len(changed)), flush=True)
    if args.write and changed:
        _hip._TUNE = dict(table)
        old = json.load(open(_hip.DEFAULTS_PATH))
        n = _hip.save_tune_defaults(note=old.get('note', '') + '; %d detect entries re-decided in the timed step by tools/tune_instep.py --detect' % len(changed))
        print('wrote %s (%d entries)' % (_hip.DEFAULTS_PATH, n))


if __name__ == '__main__':
    main()
