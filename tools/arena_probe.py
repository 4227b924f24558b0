#!/usr/bin/env python
"""Multi-scale training steps over the size schedule with the activation arena on / off and with / without the up-front reservation:
reserved memory and first-visit cost per size.    python tools/arena_probe.py [reserve|noreserve] [sizes]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import bench_data, train as y2train, utils

mode = sys.argv[1] if len(sys.argv) > 1 else 'reserve'
sizes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '320,352,384,416,448,480,512,544,576,608').split(',')]
dev = torch.device('cuda:0')
B, C = 64, 80
inf, anchors = bench_data.build_model(C, dev, 'darknet')
inf.train()
opt = utils.optim.SGD(inf.parameters(), 1e-3, momentum=0.9)
gib = lambda: round(torch.cuda.memory_reserved() / 2.0 ** 30, 2)
agib = lambda: round(torch.cuda.memory_allocated() / 2.0 ** 30, 2)


def snapshot():
    """GiB by (pool id, stream, block state) over the allocator's segments."""
    acc = {}
    for seg in torch.cuda.memory_snapshot():
        for b in seg['blocks']:
            k = '%s/%s/%s' % (seg.get('segment_pool_id'), seg.get('stream'), b['state'])
            acc[k] = acc.get(k, 0) + b['size']
    return {k: round(v / 2.0 ** 30, 2) for k, v in sorted(acc.items()) if v > (64 << 20)}



def batch(S):
    d = {k: v.to(dev) for k, v in bench_data.labels(B, S, C, seed=2 + S).items()}
    d['tensor'] = bench_data.images(B, S, seed=11 + S).to(dev)
    return d


out = {'mode': mode, 'arena': y2train.ARENA, 'reserved_gib_start': gib()}
if mode == 'reserve':
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    big = batch(max(sizes))
    got = y2train.reserve(inf, big, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
    torch.cuda.synchronize()
    out['reserve_ms'] = round((time.perf_counter() - t0) * 1e3, 1)
    out['reserved_gib_after_reserve'] = gib()
    out['allocated_gib_after_reserve'] = agib()
    out['snapshot_after_reserve'] = snapshot()
    del big
rows = []
for S in sizes:
    d = batch(S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(4):
        y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t1) / 4
    runner = inf.__dict__['_y2_step_runner']
    rows.append({'size': S, 'first_visit_ms': round(((t1 - t0) - 5 * steady) * 1e3, 1), 'ms_per_step': round(steady * 1e3, 2), 'reserved_gib': gib(), 'allocated_gib': agib(), 'snapshot': snapshot(), 'last': runner.last,
                 'loss': float(y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)['loss_total'])})
    del d
out['per_size'] = rows
out['captures'] = runner.captures
out['eager_only'] = [str(s) for s in runner.eager_only]
# second pass: every size replays, nothing grows
before = gib()
for S in sizes:
    d = batch(S)
    for _ in range(2):
        y2train.iterate(inf, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
    del d
torch.cuda.synchronize()
out['reserved_gib_second_pass'] = [before, gib()]
print(json.dumps(out))
