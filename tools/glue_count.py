#!/usr/bin/env python
"""Launches per training step split into the library's kernels and everything else (torch / rocclr glue), plus the host enqueue time
per step, for Darknet-19 VOC-20 at a given size / batch (defaults: the multi-scale schedule's smallest and the bench size).

    python tools/glue_count.py [--sizes 320,416] [--batch 64] [--dp]     (--dp: wrap in DataParallelRCCL at world size 1)

Kernel names come from torch.profiler (roctracer); a launch counts as "repo" when its name is a kernel of libyolo2_hip.so
(anonymous-namespace kernels of csrc/*.hip)."""
import argparse, collections, json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import bench_data, train as y2train, utils

ap = argparse.ArgumentParser()
ap.add_argument('--sizes', default='320,416')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--dp', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
REPO = re.compile(r'(conv|wino|bn_|opt_|loss_|decode|prep_weights|pack_weight|multi_kernel|small_|f64_to_f32|colsum|colstats|det_|maxpool|nchw|nms|iou|rowmax|compact|expand)')
if args.dp:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
out = {}
for S in [int(v) for v in args.sizes.split(',')]:
    inf, anchors = bench_data.build_model(20, dev, 'darknet')
    inf.train()
    m = y2train.DataParallelRCCL(inf) if args.dp else inf
    opt = utils.optim.SGD(m.parameters(), 1e-3, momentum=0.9)
    data = {k: v.to(dev) for k, v in bench_data.labels(args.batch, S, 20, seed=2).items()}
    data['tensor'] = bench_data.images(args.batch, S, seed=11).to(dev)
    step = lambda: y2train.iterate(m, opt, data, bench_data.HPARAM, 0.6, anchors)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(6):
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize()
        host.append((t1 - t0) * 1e3); total.append((time.perf_counter() - t0) * 1e3)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    names = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            names[e.name] += 1
    repo = {k: v / 2.0 for k, v in names.items() if REPO.search(k) and 'at::native' not in k and 'rocclr' not in k}
    glue = {k: v / 2.0 for k, v in names.items() if k not in repo}
    out[S] = dict(repo_launches_per_step=sum(repo.values()), glue_launches_per_step=sum(glue.values()), host_enqueue_ms=round(sorted(host)[len(host) // 2], 2),
                  step_ms=round(sorted(total)[len(total) // 2], 2), glue={k[:90]: v for k, v in sorted(glue.items(), key=lambda kv: -kv[1])})
    print('S=%d B=%d%s: repo launches/step %.1f, glue launches/step %.1f, host enqueue %.2f ms of a %.2f ms step' %
          (S, args.batch, ' (DP wrapper, world 1)' if args.dp else '', out[S]['repo_launches_per_step'], out[S]['glue_launches_per_step'], out[S]['host_enqueue_ms'], out[S]['step_ms']))
    for k, v in list(out[S]['glue'].items())[:40]:
        print('    %5.1f  %s' % (v, k))
    del inf, m, opt
    torch.cuda.empty_cache()
print(json.dumps(out))
