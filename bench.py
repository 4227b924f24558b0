#!/usr/bin/env python
"""bench.py — images/sec of the Darknet-19 YOLOv2 hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S] [--mode detect|forward]

N = 1 workload = BASELINE.json configs[1]: Darknet-19 YOLOv2 416x416 batch-32 inference (conv stack + decode +
visibility filter + NMS), synthetic images, random-init weights (seeded), inputs resident in HBM before the timed
region.  For N > 1 (launched by torch.distributed.run, one process per GPU) inference is "replicas only": every rank
runs the same per-GPU batch, no data-path collective; the barrier and max-over-ranks timing stay.

Prints ONE JSON line on rank 0 with the contract fields plus:
  roofline     — conv_fwd_dma_kernel family (the dominant kernel: 99% of the FLOPs): algorithmic conv FLOPs of its 22
                 launches per step / their duration, measured live with one HIP event pair per step on the launch stream
                 inside the timed region; peak = 157.3 TFLOP/s (fp32-input MFMA, MI355X_MICROARCH.md).
  cpu_baseline — the CPU oracle (port of the reference path: torch-CPU conv stack + numpy decode/filter/NMS) timed on
                 this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
APP = os.path.join(ROOT, 'yolo2-pytorch_amd')
for p in (ROOT, APP):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3


def build_model(num_cls, dev, arch='darknet'):
    import configparser

    import model
    import model.yolo2
    from oracle import darknet as odark
    from oracle import synth
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    # SURVEY.md 8d weights: seed 0, kaiming conv, randomised BN buffers; head scaled so exp(size_norm) stays finite
    if arch != 'darknet':
        import model.resnet
        from oracle import resnet as ores
        cfg.read_dict({'model': {'pretrained': '0'}})
        sd = ores.init_state_dict(arch, 5, num_cls, seed=0, head_scale=0.25)
        dnn = getattr(model.resnet, arch)(model.ConfigChannels(cfg, sd), anchors, num_cls)
        dnn.load_state_dict(sd, strict=False)
        return model.Inference(cfg, dnn, anchors).to(dev).eval(), anchors, sd
    sd = odark.init_state_dict(5, num_cls, seed=0, head_scale=1 / 40.0)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, num_cls)
    dnn.load_state_dict(sd, strict=False)
    inf = model.Inference(cfg, dnn, anchors).to(dev).eval()
    return inf, anchors, sd


def cpu_baseline(sd, anchors, size, sample):
    """Oracle (port) on the host cores: conv stack + decode + filter + NMS on `sample` images."""
    import numpy as np
    from oracle import darknet as odark
    from oracle import detect as odet
    from oracle import head as ohead
    from oracle import synth
    cores = os.cpu_count() or 1
    x = synth.images(min(sample, 16), size, seed=1).repeat((sample + 15) // 16, 1, 1, 1)[:sample]
    with torch.no_grad():
        # pick the thread count that serves the oracle best on this host (oneDNN degrades when oversubscribed)
        best = (1e30, cores)
        for th in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}):
            torch.set_num_threads(th)
            odark.forward(x[:2], sd)  # warm-up
            t0 = time.perf_counter()
            odark.forward(x[:2], sd)
            best = min(best, (time.perf_counter() - t0, th))
        torch.set_num_threads(best[1])
        t0 = time.perf_counter()
        for i0 in range(0, sample, 16):
            xb = x[i0:i0 + 16]
            feat = odark.forward(xb, sd)
            pred = ohead.decode(feat, anchors)
            B = feat.shape[0]
            prob = torch.softmax(pred['logits'], -1).view(B, -1, pred['logits'].shape[-1]).numpy()
            iou = pred['iou'].reshape(B, -1).numpy()
            mn, mx = pred['yx_min'].reshape(B, -1, 2).numpy(), pred['yx_max'].reshape(B, -1, 2).numpy()
            for b in range(B):
                odet.postprocess(iou[b], mn[b], mx[b], prob[b], fix=True)
        dt = time.perf_counter() - t0
    return {'value': round(sample / dt, 3), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d synthetic %dx%d images in batches of 16, oracle conv stack (torch-CPU fp32, best of %d/%d/%d/%d threads) + decode + filter(fix=1) + NMS, %.1f s' % (sample, size, size, cores, cores // 2, cores // 4, cores // 8, dt)}


def train_leg(args, dev, world, rank, barrier):
    """BASELINE configs[2]: Darknet-19 VOC-20 training step at 416x416, per-GPU batch 64: forward (batch-stat BN) + region
    loss + backward + SGD(lr 1e-3, momentum 0.9) (quick_start.sh:71).  N > 1: data parallel, gradients averaged with bucketed
    RCCL all-reduce overlapped with backward (train.DataParallelRCCL), weak scaling."""
    import train as y2train
    from oracle import loss as oloss
    from oracle import synth
    inf, anchors, sd = build_model(args.classes, dev, args.model)
    del sd
    inf.train()
    wrapped = y2train.ensure_model(inf)
    import utils
    opt = utils.optim.SGD(wrapped.parameters(), 1e-3, momentum=0.9)      # fused multi-tensor step (y2_opt_sgd), torch.optim.SGD semantics
    B, S = args.train_batch, args.size
    data = {k: v.to(dev) for k, v in synth.labels(B, S, args.classes, seed=2 + rank).items()}
    data['tensor'] = synth.images(B, S, seed=11 + rank).to(dev)

    def step():
        return y2train.iterate(wrapped, opt, data, oloss.HPARAM, 0.6, anchors)

    for _ in range(2):
        r = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        r = step()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    per_img = 87.78e9 * (S / 416.0) ** 2 if args.model == 'darknet' else 3 * 60.85e9 * (S / 608.0) ** 2   # fwd + wgrad + dgrad
    flops = per_img * B * args.train_steps * world
    out = {'metric': 'images/sec (%dx%d) train, %s YOLOv2 %d classes' % (S, S, 'Darknet-19' if args.model == 'darknet' else args.model, args.classes), 'value': round(B * args.train_steps * world / dt, 2), 'unit': 'images/sec',
           'ms_per_step': round(dt / args.train_steps * 1e3, 3), 'steps': args.train_steps, 'per_gpu_batch': B, 'global_batch': B * world,
           'parallelism': 'dp%d (RCCL all-reduce, bucketed, overlapped with backward)' % world if world > 1 else 'single GPU',
           'optimizer': 'utils.optim.SGD(lr=1e-3, momentum=0.9): fused multi-tensor HIP kernel, torch.optim.SGD semantics', 'loss_total': float(r['loss_total'].detach()),
           'conv_tflops': round(flops / dt / 1e12 / world, 2), 'conv_frac_of_fp32_mfma_peak': round(flops / dt / 1e12 / world / PEAK_FP32_MFMA_TFLOPS, 4)}
    del wrapped, opt, inf
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--mode', default='detect', choices=['detect', 'forward'])
    ap.add_argument('--model', default='darknet', help="darknet (default, BASELINE configs[1]) or a model.resnet plugin name, e.g. resnet50 (configs[4] forward: --size 608 --classes 80)")
    ap.add_argument('--no-direct-leg', action='store_true', help='skip the extra Winograd-off measurement (roofline.direct_only)')
    ap.add_argument('--no-graph', action='store_true', help='launch the detect step eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--train-steps', type=int, default=6, help='extra leg: timed training steps (fwd + region loss + bwd + SGD), 0 = skip')
    ap.add_argument('--settle', type=float, default=2.0, help='seconds of idle between the inference legs and the training leg (outside every timed region)')
    ap.add_argument('--train-batch', type=int, default=64, help='per-GPU batch of the training leg (BASELINE configs[2])')
    ap.add_argument('--cpu-sample', type=int, default=192, help='images for the CPU baseline (0 = skip)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import train as y2train
        y2train.init_distributed()                      # backend "nccl" == RCCL on ROCm, one process per GPU
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    import detect
    from oracle import synth
    inf, anchors, sd = build_model(args.classes, dev, args.model)
    dnn = inf.dnn
    if args.model != 'darknet':
        args.cpu_sample = 0
    x = synth.images(args.batch, args.size, seed=1 + rank).to(dev)   # resident in HBM before timing

    def step():
        with torch.no_grad():
            feat = dnn.forward_nhwc(x)
            if args.mode == 'detect':
                return detect.detect_batch(feat, anchors, fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
            return feat

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import _hip

    def conv_chain_rate(prof):
        """(executed TF/s, direct-equivalent TF/s, chain ms/step, conv0 ms/step, Winograd layer count, flops/step, executed flops/step)"""
        fl = fl_exec = ms = ms0 = 0.0
        n_launch = n_wino = 0
        for rec in prof:
            name, flops, e0, e1 = rec[:4]
            d = e0.elapsed_time(e1)
            if name.startswith('conv_fwd'):
                fl += flops
                fl_exec += rec[4] if len(rec) > 4 else flops
                n_wino = rec[5] if len(rec) > 5 else 0
                ms += d
                n_launch += 1
            else:
                ms0 += d
        n = max(1, n_launch)
        rate = lambda f: f / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return rate(fl_exec), rate(fl), ms / n, ms0 / n, n_wino, fl / n, fl_exec / n

    def measure(steps, warmup):
        """W untimed warm-up steps, a roofline leg (eager steps, one HIP event pair around the conv chain each: events cannot
        be queried inside a graph), then EXACTLY `steps` timed steps (hipGraph replay) between barrier + synchronize."""
        for _ in range(warmup):
            step()
        barrier()
        dnn.profile = []
        for _ in range(min(steps, 10)):
            step()
        barrier()
        prof, dnn.profile = dnn.profile, None
        graphed = None
        if not args.no_graph and args.mode == 'detect':
            try:
                graphed = detect.GraphedDetector(dnn, anchors, x, fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
            except Exception as e:   # capture not possible -> eager launches
                print('hipGraph capture failed (%s); eager launches' % e, file=sys.stderr)
                graphed = None
        run = (lambda: graphed.run()) if graphed is not None else step
        for _ in range(2):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        host_dt = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), host_dt, prof, graphed is not None

    dt, host_dt, prof, was_graphed = measure(args.steps, args.warmup)
    executed, direct_equiv, chain_ms, conv0_ms, n_wino, fl_step, fl_exec_step = conv_chain_rate(prof)

    # second leg, reported beside the metric: the same step with the Winograd algorithm disabled = pure implicit-GEMM
    # convolutions (executed == algorithmic multiply-adds), the conv-MFMA roofline fraction north_star asks for
    direct_leg = None
    if _hip.WINOGRAD and args.model == 'darknet' and not args.no_direct_leg:      # rank-independent condition: every rank runs the same barriers
        _hip.WINOGRAD = False
        dnn._plan_cache = None
        try:
            ddt, _, dprof, _ = measure(min(args.steps, 10), 2)
            dex, _, dchain, _, _, _, _ = conv_chain_rate(dprof)
            direct_leg = {'images_per_sec': round(args.batch * min(args.steps, 10) * world / ddt, 2), 'achieved': round(dex, 2),
                          'frac': round(dex / PEAK_FP32_MFMA_TFLOPS, 4), 'conv_chain_ms_per_step': round(dchain, 4),
                          'note': 'Y2_WINOGRAD=0: every 3x3 layer through the implicit-GEMM MFMA kernel'}
        except Exception as e:
            import traceback
            traceback.print_exc()
            direct_leg = {'error': '%s: %s' % (type(e).__name__, e)}
        finally:
            _hip.WINOGRAD = True
            dnn._plan_cache = None

    train_out = None
    if args.train_steps > 0:
        torch.cuda.synchronize()
        time.sleep(args.settle)          # untimed pause between the inference legs and the training leg
        try:
            train_out = train_leg(args, dev, world, rank, barrier)
        except Exception as e:           # the extra leg must not take the headline metric down with it
            import traceback
            traceback.print_exc()
            train_out = {'error': '%s: %s' % (type(e).__name__, e)}
    traffic = None
    try:   # HBM-side bytes per step of the same kernel family from the committed rocprofv3 PMC passes (separate runs)
        import glob
        tf = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_detect_b32_traffic.json')) if '_direct_' not in os.path.basename(f))
        if tf and args.batch == 32 and args.size == 416 and args.model == 'darknet':
            traffic = json.load(open(tf[-1]))['traffic_bytes_per_step']
        tfd = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_direct_detect_b32_traffic.json')))
        if direct_leg is not None and 'error' not in direct_leg and tfd and args.batch == 32 and args.size == 416 and args.model == 'darknet':
            direct_leg['traffic'] = json.load(open(tfd[-1]))['traffic_bytes_per_step']
    except Exception:
        traffic = None
    if rank == 0:
        images = args.batch * args.steps * world
        out = {
            'metric': 'images/sec (416x416) detect, Darknet-19 YOLOv2',
            'value': round(images / dt, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'launch': 'hipGraph replay' if was_graphed else 'eager', 'host_ms_per_step': round(host_dt / args.steps * 1e3, 4),
            'config': {'model': args.model, 'workload': 'Darknet-19 YOLOv2 %dx%d batch-%d/GPU inference: conv stack + decode + filter + NMS (BASELINE configs[1])'
                                   % (args.size, args.size, args.batch) if args.mode == 'detect' else
                                   'Darknet-19 YOLOv2 %dx%d batch-%d/GPU conv stack only' % (args.size, args.size, args.batch),
                       'classes': args.classes, 'global_batch': args.batch * world, 'parallelism': 'replicas x%d (no collective)' % world,
                       'weights': 'random-init seed 0'},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_fwd_dma_kernel family (fp32 MFMA GEMMs: implicit-GEMM direct convs + the grouped GEMMs of the Winograd layers, with their transform kernels; the 22-layer chain timed as one event pair per step, inter-launch gaps included)',
                         'achieved': round(direct_equiv, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(direct_equiv / PEAK_FP32_MFMA_TFLOPS, 4),
                         'definition': 'ALGORITHMIC conv FLOPs (SURVEY.md 8d: 2*Cin*Cout*k*k*H*W per layer, 29.061 GFLOP/img) / measured chain time; it may exceed 1 because %d layers run Winograd F(2x2,3x3), which executes 16/36 of those multiply-adds' % n_wino,
                         'mfma_executed_tflops': round(executed, 2), 'mfma_utilisation': round(executed / PEAK_FP32_MFMA_TFLOPS, 4),
                         'winograd_layers': n_wino, 'executed_flops_per_step': fl_exec_step,
                         'flops_per_step': fl_step, 'ms_per_step': round(chain_ms, 4),
                         'conv0_ms_per_step': round(conv0_ms, 4), 'traffic': traffic,
                         'traffic_note': 'bytes per step (conv chain) from rocprofv3 PMC FETCH_SIZE(x2, gfx950 correction)+WRITE_SIZE, profiles/; L2 memory-side requests incl. Infinity-Cache hits',
                         'direct_only': direct_leg},
        }
        if train_out is not None:
            out['train'] = train_out
        if world == 1 and args.cpu_sample > 0:
            try:
                out['cpu_baseline'] = cpu_baseline(sd, anchors, args.size, args.cpu_sample)
            except Exception as e:
                out['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
