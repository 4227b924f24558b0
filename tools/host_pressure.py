#!/usr/bin/env python
"""Rehearsal of eight ranks sharing one host, on one GPU: the training step's HOST cost under CPU confinement.

With one process per GPU the eight ranks of a node share its cores; what the GPU-bound single-rank run hides (the launch queue absorbs
the host's lead) shows when a rank owns an eighth of the cores and seven other interpreters are busy beside it.  For each input size
this tool runs the batch-64 training step (train.iterate: forward + region loss + backward + fused SGD) in its three launch modes -
autograd (one ctypes launch per kernel under torch.autograd), plan (same launches, no autograd), graph (captured hipGraph replay, linear), graph+fork (weight gradients on a second branch) - in
three host settings:

  free        all cores, nobody else
  pinned      this process confined to cpu_count/8 cores; 7 busy Python peers on the OTHER cores (ranks pinned like bench.py --gpus 8 does)
  crowded     this process AND the 7 busy peers confined to the same cpu_count/8 cores (unpinned ranks at their worst)

and prints step time, the host time to issue a step with the GPU idle, and their ratio.  Output: one JSON line per (size, mode, setting).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

BUSY = 'import time\nx = 0\nwhile True:\n    for i in range(100000):\n        x += i * i\n'


def peers(n, cores):
    procs = []
    for i in range(n):
        p = subprocess.Popen([sys.executable, '-c', BUSY])
        if cores is not None:
            os.sched_setaffinity(p.pid, cores)
        procs.append(p)
    return procs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='320,416,608')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--settings', default='free,pinned,crowded')
    args = ap.parse_args()
    import torch

    import bench_data
    import train as y2train
    import utils
    dev = torch.device('cuda', 0)
    all_cores = sorted(os.sched_getaffinity(0))
    share = all_cores[:max(1, len(all_cores) // 8)]
    others = all_cores[len(share):] or all_cores
    torch.set_num_threads(max(1, len(share)))
    rows = []
    for S in [int(v) for v in args.sizes.split(',')]:
        d = {k: v.to(dev) for k, v in bench_data.labels(args.batch, S, 20, seed=2).items()}
        d['tensor'] = bench_data.images(args.batch, S, seed=11).to(dev)
        from model import train_graph
        for mode, (plan, graph, fork) in (('autograd', (False, False, False)), ('plan', (True, False, False)), ('graph', (True, True, False)), ('graph+fork', (True, True, True))):
            # graph = ONE linear hipGraph per step (what the data-parallel wrapper replays); graph+fork = the weight gradients on a second branch (single-process default, round 5)
            y2train.PLAN, y2train.GRAPH, train_graph.GRAPH_FORK = plan, graph, fork
            inf, anchors = bench_data.build_model(20, dev, 'darknet')
            inf.train()
            opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)

            def step():
                return y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            for setting in args.settings.split(','):
                procs = []
                try:
                    if setting == 'pinned':
                        os.sched_setaffinity(0, share)
                        procs = peers(7, others)
                    elif setting == 'crowded':
                        os.sched_setaffinity(0, share)
                        procs = peers(7, share)
                    else:
                        os.sched_setaffinity(0, all_cores)
                    time.sleep(0.5)
                    step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / args.steps * 1e3
                    issue = []
                    for _ in range(6):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        step()
                        issue.append((time.perf_counter() - t0) * 1e3)
                    torch.cuda.synchronize()
                    issue = sorted(issue)[len(issue) // 2]
                finally:
                    for p in procs:
                        p.kill()
                    for p in procs:
                        p.wait()
                    os.sched_setaffinity(0, all_cores)
                row = dict(size=S, batch=args.batch, mode=mode, setting=setting, cores=len(share) if setting != 'free' else len(all_cores), ms_per_step=round(ms, 3),
                           host_issue_ms=round(issue, 3), host_over_step=round(issue / ms, 3), images_per_sec=round(args.batch / ms * 1e3, 1))
                rows.append(row)
                print(json.dumps(row), flush=True)
            del inf, opt
            torch.cuda.empty_cache()
    y2train.PLAN = y2train.GRAPH = True
    train_graph.GRAPH_FORK = 'auto'


if __name__ == '__main__':
    main()
