#!/usr/bin/env python
"""bench.py — images/sec of the Darknet-19 YOLOv2 hot path on MI355X (BASELINE.json metric: train + detect at 1/2/4/8 GPUs).

    python bench.py [--gpus N] [--steps K] [--warmup W]        # N > 1: re-launches itself as N ranks (one process per GPU)

Launch.  With WORLD_SIZE unset and --gpus N > 1 the script re-executes itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; launched by the driver through torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  `ranks_seen` in the output is an RCCL all-reduce of ones.

Legs (every timed region: W untimed warm-up steps, barrier + synchronize, EXACTLY K steps, barrier + synchronize, MAX over ranks):
  detect   BASELINE configs[1]: 416x416 batch-32 inference per GPU = conv stack + decode + visibility filter + NMS, inputs
           resident in HBM, K hipGraph replays rotating over 4 different resident batches.  N > 1: replicas only (no collective).
  train    BASELINE configs[2]: VOC-20 batch-64 per GPU: forward (batch-stat BN) + region loss + backward + fused SGD
           (lr 1e-3, momentum 0.9; quick_start.sh:71).  N > 1: data parallel through train.ensure_model (bucketed RCCL
           all-reduce overlapped with backward, positive-count all-reduce), weak scaling; the same step WITHOUT the wrapper is
           timed beside it on every rank (`train.single_gpu_images_per_sec`: the denominator of the DP scaling efficiency).
  Headline `value`: the batch-64 training step (BASELINE configs[2]) at EVERY N - one workload for the whole scaling curve (north_star:
           ">= 6x DP scaling 1 -> 8"); detect is reported beside it (`roofline.detect_*` / `summary.detect_*`).  --headline overrides.

Roofline (N = 1, measured in this run with the library's per-kernel HIP-event hooks, y2_prof_*, on the launch stream):
  roofline         dominant MFMA kernel TEMPLATE of the headline step: executed multiply-add FLOPs per step of all its launches / its time per step against
                   the fp32-input MFMA peak (157.3 TFLOP/s).  `frac`: time from the committed rocprofv3 trace of the timed schedule (profiles/*_traffic.json,
                   same kernel sources, same launches per step; else the event-hook figure, see `frac_source`); `frac_uncontended`: HIP event pairs per launch
                   in this run (eager, single stream); `step_frac_executed` / `step_frac_direct_equiv`: the whole timed step in executed / SURVEY 8d FLOPs.
  conv3x3_b64      north_star's target quantity: the 3x3 convolutions at batch 64, per-layer event pairs, Winograd on and off.
  train.roofline   the same per-kernel table for the training step (executed FLOPs of fprop / dgrad / wgrad kernels).
  cpu_baseline     the CPU oracle (port of the reference path) on this box's host cores, bounded sample, rank 0 at N = 1 only: batch-16 chunks, the
                   reference's own one-image-per-call shape (BASELINE configs[0]), one batch-8 training step, NMS at 200 candidates.
  latency          (round 4) configs[0] on the GPU: batch 1 / batch 8, one hipGraph, serial replays, with plan, launch count and both floors.
  resnet50_608     (round 4) BASELINE configs[4] per GPU: ResNet-50 plugin, 608x608, COCO-80, batch 32: detect, train, per-kernel table.
  multiscale       BASELINE configs[3] per GPU: COCO-80, sizes 320..608, resize every 10 batches.
  The training legs go through train.iterate: a captured hipGraph per input shape (model.train_graph.StepPlan); `host_issue_ms_per_step` is the host
  time to issue one step with the GPU idle.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
APP = os.path.join(ROOT, 'yolo2-pytorch_amd')
for p in (ROOT, APP):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32, dense, MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0         # v_mfma_f32_32x32x16_bf16, dense (same guide)
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0     # opt-in split mode: six bf16 plane products per fp32-equivalent multiply-add (csrc/gemm_split.hip)


def kernel_peak(name):
    return PEAK_BF16_MFMA_TFLOPS / 3.0 if name.startswith('gemm_split_f16') else PEAK_SPLIT_TFLOPS if name.startswith('gemm_split') else PEAK_FP32_MFMA_TFLOPS
FLOPS_FWD_PER_IMG = 29.360e9           # sum over the 23 convs of 2*Cin*Cout*k*k*H*W at 416x416, VOC-20
FLOPS_TRAIN_PER_IMG = 87.78e9          # fwd + wgrad + dgrad (all but the first conv)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch of the detect leg (BASELINE configs[1])')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--model', default='darknet', help='darknet (BASELINE configs[1-3]), tiny, or a model.resnet plugin name (configs[4]: --model resnet50 --size 608 --classes 80)')
    ap.add_argument('--headline', default='auto', choices=['auto', 'detect', 'train'], help='auto: the batch-64 training step (BASELINE configs[2]) at EVERY N - one workload for the whole scaling curve; detect beside it')
    ap.add_argument('--train-steps', type=int, default=0, help='timed training steps (0 = --steps)')
    ap.add_argument('--train-batch', type=int, default=64, help='per-GPU batch of the train leg (BASELINE configs[2])')
    ap.add_argument('--rotate', type=int, default=4, help='number of different resident input batches the timed steps rotate over')
    ap.add_argument('--no-graph', action='store_true', help='launch the detect step eagerly instead of replaying captured hipGraphs')
    ap.add_argument('--streams', type=int, default=2, help='HIP streams the detect steps are pipelined over (2: batch i+1 starts while the last workgroups of batch i drain; 1: strictly serial)')
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--no-detect', action='store_true')
    ap.add_argument('--no-direct-leg', action='store_true', help='skip the Winograd-off measurements')
    ap.add_argument('--no-conv3', action='store_true', help='skip the batch-64 conv3x3 leg')
    ap.add_argument('--no-split-leg', action='store_true', help='skip the opt-in split-bf16 precision mode measurement (Y2_SPLIT_BF16=1: Winograd GEMMs on the bf16 matrix pipe)')
    ap.add_argument('--settle', type=float, default=2.0, help='idle seconds between the inference legs and the train leg (outside timed regions)')
    ap.add_argument('--cpu-sample', type=int, default=192, help='images for the CPU baseline (0 = skip)')
    ap.add_argument('--multiscale', action='store_true', help='only the multi-scale training leg (BASELINE configs[3]) at full length')
    ap.add_argument('--no-multiscale', action='store_true', help='skip the multi-scale training leg')
    ap.add_argument('--ms-classes', type=int, default=80, help='classes of the multi-scale leg (COCO-80)')
    ap.add_argument('--ms-sizes', default='320,352,384,416,448,480,512,544,576,608', help='input sizes of the multi-scale schedule (config.ini:39-40: 320..608 step 32)')
    ap.add_argument('--ms-maintain', type=int, default=10, help='batches per size before the next resize (config.ini [data] maintain, utils/data.py:135-141)')
    ap.add_argument('--ms-cycles', type=int, default=1, help='timed passes over the whole size schedule')
    ap.add_argument('--no-latency', action='store_true', help='skip the batch-1 / batch-8 detect latency leg (BASELINE configs[0] twin on the GPU)')
    ap.add_argument('--no-resnet', action='store_true', help='skip the ResNet-50 608x608 COCO-80 leg (BASELINE configs[4] per GPU)')
    ap.add_argument('--resnet-batch', type=int, default=32, help='per-GPU batch of the ResNet-50 leg')
    ap.add_argument('--tables', default='', help='where the long form of the result (per-kernel tables, plans) is written (default gpurun_out/bench_full.json)')
    ap.add_argument('--dry-run', action='store_true', help='no GPU: exercise launch / rendezvous / DP wrapper / timing protocol with a stand-in CPU workload (gloo); the numbers mean nothing')
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """--gpus N without a launcher: become `torch.distributed.run` with N ranks on this node (one process per GPU)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL needs it on this driver)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------- measurement helpers
def pin_rank(local, world):
    """One process per GPU on ONE host: give every rank its own contiguous share of the cores this job may use (issue thread, RCCL proxy
    and torch's intra-op threads of a rank stay on one set of caches instead of migrating over the whole socket pair).  Y2_PIN=0: leave the
    affinity alone.  Returns the number of cores the rank ends up with."""
    try:
        have = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1
    if world <= 1 or os.environ.get('Y2_PIN', '1') == '0' or len(have) < 2 * world:
        return len(have)
    per = len(have) // world
    mine = have[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return len(have)
    return len(mine)


class Ctx(object):
    """Process-wide run context: ranks, device, barrier, max-over-ranks."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local = int(os.environ.get('LOCAL_RANK', '0'))
        self.gpu = torch.cuda.is_available() and not args.dry_run
        self.cores = pin_rank(self.local, self.world)
        if self.gpu:
            if os.environ.get('Y2_BENCH_DEVICE'):      # test rig only: several ranks on ONE GPU (with Y2_DIST_BACKEND=gloo) to exercise the N > 1 code path
                self.local = int(os.environ['Y2_BENCH_DEVICE'])
                os.environ['LOCAL_RANK'] = str(self.local)
            torch.cuda.set_device(self.local)
            self.dev = torch.device('cuda', self.local)
        else:
            self.dev = torch.device('cpu')
        if self.world > 1:
            import train as y2train
            if not self.gpu:
                os.environ.setdefault('Y2_DIST_BACKEND', 'gloo')
            y2train.init_distributed()                      # backend "nccl" == RCCL on ROCm, one process per GPU
        ones = torch.ones(1, device=self.dev)
        if self.world > 1:
            dist.all_reduce(ones)
        self.ranks_seen = int(ones.item())

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, seconds):
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def timed(self, fn, steps):
        """barrier + synchronize | EXACTLY `steps` calls | barrier + synchronize; MAX over ranks.  Returns (seconds, host seconds)."""
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        host = time.perf_counter() - t0
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), host


def release_memory():
    """End of a leg: collect (captured graphs sit in reference cycles - plan <-> tape <-> closures - and their pools stay reserved until the
    collector has run), then hand the allocator's cached blocks back to the driver."""
    import gc

    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def kernel_table(fn, steps):
    """Run fn(i) `steps` times with the library's per-kernel event hooks on (include/yolo2_hip.h: y2_prof_*): every kernel launch
    is bracketed by a HIP event pair on its launch stream.  Returns {kernel: dict(launches, ms, flops)} per STEP (averages)."""
    import ctypes

    import torch

    import _hip
    L = _hip.lib()
    torch.cuda.synchronize()
    L.y2_prof_enable(1)
    try:
        for i in range(steps):
            fn(i)
        torch.cuda.synchronize()
    finally:
        L.y2_prof_enable(0)
    name = ctypes.create_string_buffer(96)
    ms, fl = ctypes.c_float(), ctypes.c_double()
    table = {}
    for i in range(L.y2_prof_count()):
        if L.y2_prof_get(i, name, 96, ctypes.byref(ms), ctypes.byref(fl)) != 0:
            continue
        e = table.setdefault(name.value.decode(), dict(launches=0, ms=0.0, flops=0.0))
        e['launches'] += 1
        e['ms'] += ms.value
        e['flops'] += fl.value
    for e in table.values():
        e['launches'] /= float(steps)
        e['ms'] /= steps
        e['flops'] /= steps
    return table


def issue_time(ctx, step, n=6):
    """Host time to ISSUE one step with the GPU idle (median of n, synchronised before each): `host_ms_per_step` of a timed region also
    contains the time the host spends blocked on a full launch queue, which says nothing about the cost of issuing."""
    t = []
    for i in range(n):
        ctx.sync()
        t0 = time.perf_counter()
        step(i)
        t.append((time.perf_counter() - t0) * 1e3)
    ctx.sync()
    return round(sorted(t)[len(t) // 2], 3)


def top_kernels(table, min_share=0.01):
    total = sum(e['ms'] for e in table.values()) or 1.0
    rows = []
    for k, e in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
        if e['ms'] / total < min_share:
            continue
        tf = e['flops'] / (e['ms'] * 1e-3) / 1e12 if e['flops'] > 0 and e['ms'] > 0 else None
        rows.append({'kernel': k, 'launches_per_step': round(e['launches'], 2), 'ms_per_step': round(e['ms'], 4), 'share': round(e['ms'] / total, 4),
                     'avg_launch_us': round(e['ms'] / max(e['launches'], 1e-9) * 1e3, 2),
                     'executed_tflops': None if tf is None else round(tf, 2), 'frac': None if tf is None else round(tf / kernel_peak(k), 4)})
    return rows, total


def family(name):
    """'conv_fwd_dma_kernel[grouped]' -> 'conv_fwd_dma_kernel': the __global__ template a hook name belongs to (the names rocprofv3 prints)."""
    return name.split('[')[0]


def families(table):
    """Hook table grouped by kernel template: {family: dict(launches, ms, flops)} per step."""
    fam = {}
    for k, e in table.items():
        f = fam.setdefault(family(k), dict(launches=0.0, ms=0.0, flops=0.0))
        for key in ('launches', 'ms', 'flops'):
            f[key] += e[key]
    return fam


def roofline_from(table, what, trace_tag=None):
    """`roofline` object for the dominant MFMA kernel of `table` + the per-kernel list.  The dominant kernel is a kernel TEMPLATE (all its launches of a
    step, whatever job each does: the names a rocprofv3 trace prints).  `frac` is what the committed rocprofv3 trace of the timed schedule supports
    (profiles/*_traffic.json: `trace`, same kernel sources, same launches per step) when there is one - this run's executed FLOPs per step of the
    kernel / the trace's time per step of the kernel; `frac_uncontended` is this run's event-hook figure (eager single-stream launches)."""
    rows, total_ms = top_kernels(table)
    fam = families(table)
    mf = {k: e for k, e in fam.items() if e['flops'] > 0 and e['ms'] > 0}
    dom = max(mf, key=lambda k: mf[k]['ms']) if mf else None
    gemm_ms = sum(e['ms'] for e in table.values() if e['flops'] > 0)
    gemm_fl = sum(e['flops'] for e in table.values())
    out = {'bound': 'mfma', 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'what': what,
           'definition': 'executed multiply-add FLOPs of the kernel template per step (2*M*N*K of the GEMM each launch runs; Winograd layers execute 16/36 of the direct count) / its '
                         'time per step: `frac` from the committed rocprofv3 kernel trace of the timed schedule when it matches this build, `frac_uncontended` from HIP event pairs '
                         'per launch on the launch stream in this run',
           'kernel': dom, 'achieved': None, 'frac': None, 'frac_uncontended': None, 'frac_source': None,
           'all_mfma_kernels': {'executed_tflops': round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 2) if gemm_ms > 0 else None,
                                'frac': round(gemm_fl / (gemm_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if gemm_ms > 0 else None,
                                'ms_per_step': round(gemm_ms, 4), 'executed_flops_per_step': gemm_fl},
           'kernel_ms_per_step': round(total_ms, 4), 'top_kernels': rows,
           'families': {k: {'launches_per_step': round(e['launches'], 2), 'ms_per_step': round(e['ms'], 4), 'executed_flops_per_step': e['flops']} for k, e in fam.items() if e['ms'] / (total_ms or 1.0) >= 0.01}}
    if dom:
        e = mf[dom]
        live = e['flops'] / (e['ms'] * 1e-3) / 1e12
        out.update(kernel_share_of_step=round(e['ms'] / (total_ms or 1.0), 4), launches_per_step=round(e['launches'], 2),
                   frac_uncontended=round(live / kernel_peak(dom), 4), avg_launch_us_uncontended=round(e['ms'] / e['launches'] * 1e3, 2))
        tr, why = trace_family(trace_tag, dom, e['launches']) if trace_tag else (None, 'no committed trace for this workload')
        if tr is not None:
            tf = e['flops'] / (tr['us_per_step'] * 1e-6) / 1e12
            out.update(achieved=round(tf, 2), frac=round(tf / kernel_peak(dom), 4), avg_launch_us=round(tr['us_per_step'] / e['launches'], 2), frac_source=tr['source'])
        else:
            out.update(achieved=round(live, 2), frac=out['frac_uncontended'], avg_launch_us=out['avg_launch_us_uncontended'],
                       frac_source='event hooks of this run, eager single-stream launches (%s)' % why)
    return out


def static_traffic(tag):
    """HBM-side bytes per step from the committed rocprofv3 PMC passes (separate profiling runs, tools/gpu_profile.sh): NOT measured
    in this run, so it is reported with its source, never as a live number."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_traffic.json' % tag)))
    if not files:
        return None, None
    try:
        import _hip
        d = json.load(open(files[-1]))
        rel = os.path.relpath(files[-1], ROOT)
        if d.get('kernels') != _hip.kernel_hash():
            # a PMC profile of OTHER kernels says nothing about this build: no number rather than a stale one
            return None, 'stale: %s was taken on kernel sources %s, this build is %s (regenerate with tools/gpu_profile.sh)' % (rel, d.get('kernels'), _hip.kernel_hash())
        return d['traffic_bytes_per_step'], 'static: %s (same kernel sources %s)' % (rel, d.get('kernels'))
    except Exception:
        return None, None


def static_extra(tag, key):
    """Another member of the committed profile record static_traffic() reads (same hash gate)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_traffic.json' % tag)))
    try:
        import _hip
        d = json.load(open(files[-1]))
        return d.get(key) if d.get('kernels') == _hip.kernel_hash() else None
    except Exception:
        return None


def trace_family(tag, fam, live_launches_per_step):
    """Time per step of one kernel template in the committed rocprofv3 kernel trace of the timed schedule (profiles/*_<tag>_traffic.json: `trace`,
    tools/trace_families.py).  Accepted only for the same kernel sources AND the same number of launches of that template per step as this run
    (a changed algorithm table or grouping makes the two describe different launch sets).  Returns (dict(us_per_step, source) | None, why not)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_traffic.json' % tag)))
    if not files:
        return None, 'no committed trace'
    try:
        import _hip
        d = json.load(open(files[-1]))
        rel = os.path.relpath(files[-1], ROOT)
        if d.get('kernels') != _hip.kernel_hash():
            return None, 'committed trace %s is of other kernel sources' % rel
        t = d.get('trace') or {}
        f = (t.get('families') or {}).get(fam)
        if not f or not t.get('steps'):
            return None, '%s has no trace row for %s' % (rel, fam)
        lps = f['calls'] / float(t['steps'])
        if abs(lps - live_launches_per_step) > 0.02 * max(live_launches_per_step, 1.0):
            return None, '%s: %s is launched %.2f x per step in the trace, %.2f x in this run' % (rel, fam, lps, live_launches_per_step)
        return {'us_per_step': f['total_us'] / float(t['steps']), 'source': 'rocprofv3 kernel trace of the timed schedule: %s (same kernel sources %s, %.0f launches per step)' % (rel, d.get('kernels'), lps)}, None
    except Exception as e:
        return None, '%s: %s' % (type(e).__name__, e)


# ---------------------------------------------------------------------------------------------------- detect leg
def detect_leg(args, ctx):
    import torch

    import _hip
    import bench_data
    import detect
    inf, anchors = bench_data.build_model(args.classes, ctx.dev, args.model)
    dnn = inf.dnn
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
    xs = [bench_data.images(args.batch, args.size, seed=1 + ctx.rank * 16 + i).to(ctx.dev) for i in range(max(1, args.rotate))]   # resident in HBM

    def eager(i):
        with torch.no_grad():
            return detect.detect_batch(dnn.forward_nhwc(xs[i % len(xs)]), anchors, **kw)

    def measure(steps, warmup, want_table, want_streams=None):
        for i in range(warmup):
            eager(i)
        ctx.sync()
        if ctx.world > 1 and not getattr(measure, 'synced', False):
            # replicas: every rank measured its own algorithm choices during the warm-up; near-ties resolve differently and the job's rate
            # is the slowest replica's - adopt rank 0's table (the plans are rebuilt on it by the next call)
            import train as y2train
            measure.synced = True
            measure.tune_synced = y2train.sync_tune(ctx.dev)
            eager(0)
            ctx.sync()
        table = kernel_table(eager, min(steps, 8)) if want_table else None
        runs = None
        nstreams = max(1, min(args.streams if want_streams is None else want_streams, len(xs))) if (not args.no_graph and args.model == 'darknet') else 1
        if not args.no_graph:
            try:      # one captured graph per resident batch; graph i runs in buffer slot i % nstreams on stream i % nstreams: graphs of one
                      # slot share intermediate buffers and are serial on their stream, different slots overlap
                runs = [detect.GraphedDetector(dnn, anchors, x, static_input=True, slot=i % nstreams, **kw) if nstreams > 1 else
                        detect.GraphedDetector(dnn, anchors, x, static_input=True, **kw) for i, x in enumerate(xs)]
            except Exception as e:
                print('hipGraph capture failed (%s); eager launches' % e, file=sys.stderr)
                runs, nstreams = None, 1
        streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else None

        def replay(i):
            g = i % len(runs)
            if streams is None:
                runs[g].run()
            else:
                with torch.cuda.stream(streams[g % nstreams]):
                    runs[g].run()
        fn = replay if runs is not None else eager
        measure.nstreams, measure.runs = nstreams, runs
        for i in range(2 * len(xs)):
            fn(i)
        dt, host = ctx.timed(fn, steps)
        return dt, host, table, runs is not None

    dt, host, table, graphed = measure(args.steps, args.warmup, ctx.world == 1)
    pipelined = getattr(measure, 'nstreams', 1)
    serial_dt = None
    if pipelined > 1 and ctx.world == 1:          # the same graphs strictly one after the other: the latency of a step, reported beside the throughput
        serial_steps = min(args.steps, 24)
        serial_dt = measure(serial_steps, 0, False, want_streams=1)[0] / serial_steps
        measure.nstreams = pipelined
    d2h_dt = None
    if ctx.world == 1 and getattr(measure, 'runs', None):
        # the reference hands the survivors to the host (utils/postprocess.py:34-49 returns a Python list; detect.py:69-79 indexes with it): the same replays, each
        # followed by the per-class expansion (y2_expand_classes) and the copy of the batch's detections to host memory (one copy per result buffer, one synchronisation)
        runs, nd = measure.runs, min(args.steps, 24)

        def to_host(i):
            return detect.postprocess_batch(runs[i % len(runs)].run(), fix=True, threshold_cls=kw['threshold_cls'], to_host=True)
        to_host(0)
        d2h_dt = ctx.timed(to_host, nd)[0] / nd
    images = args.batch * args.steps * ctx.world
    out = {'images_per_sec': round(images / dt, 2), 'ms_per_step': round(dt / args.steps * 1e3, 4), 'steps': args.steps,
           'host_ms_per_step': round(host / args.steps * 1e3, 4),
           'launch': ('hipGraph replay, steps pipelined over %d streams (private buffers per stream)' % measure.nstreams if getattr(measure, 'nstreams', 1) > 1 else 'hipGraph replay') if graphed else 'eager',
           'resident_batches_rotated': len(xs), 'per_gpu_batch': args.batch, 'streams': pipelined,
           'serial_ms_per_step': None if serial_dt is None else round(serial_dt * 1e3, 4), 'serial_images_per_sec': None if serial_dt is None else round(args.batch / serial_dt, 2),
           'to_host_ms_per_step': None if d2h_dt is None else round(d2h_dt * 1e3, 4), 'to_host_images_per_sec': None if d2h_dt is None else round(args.batch / d2h_dt, 2),
           'parallelism': 'replicas x%d (no collective)' % ctx.world if ctx.world > 1 else 'single GPU'}
    if ctx.world > 1:
        out['autotune_choices_synced'] = getattr(measure, 'tune_synced', None)
    roof = None
    if table is not None:
        std = args.batch == 32 and args.size == 416 and args.model == 'darknet'
        roof = roofline_from(table, 'detect step, batch %d (the timed step = hipGraph replays of these kernels pipelined over %d streams)' % (args.batch, pipelined), trace_tag='detect_b32' if std else None)
        plan = dnn._plan_cache[1] if dnn._plan_cache else None
        if plan is not None and 'flops_executed' in plan:
            conv_ms = sum(e['ms'] for k, e in table.items() if k.startswith(('conv', 'wino')))
            alg = plan['flops'] + plan['flops0']
            exe = plan['flops_executed'] + plan['flops0']
            roof['conv_chain'] = {'ms_per_step': round(conv_ms, 4), 'executed_tflops': round(exe / conv_ms / 1e9, 2), 'frac': round(exe / conv_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                                  'direct_equiv_tflops': round(alg / conv_ms / 1e9, 2),
                                  'direct_equiv_note': 'ALGORITHMIC conv FLOPs (SURVEY.md 8d, 2*Cin*Cout*k*k*H*W) / kernel time: what a direct convolution would have to sustain; not a roofline fraction',
                                  'winograd_layers': int(sum(plan['algos'])), 'flops_per_step': alg, 'executed_flops_per_step': exe}
            # the whole TIMED step (conv chain + decode + filter + NMS, pipelined) against the fp32-MFMA peak: executed multiply-adds, and SURVEY.md 8d's
            # algorithmic (direct-convolution) FLOPs - the latter exceeds 1 where Winograd layers execute 16/36 of the direct count
            roof['step_frac_executed'] = round(exe / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
            roof['step_frac_direct_equiv'] = round(alg / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        tr, src = static_traffic('detect_b32') if std else (None, None)
        roof['traffic'], roof['traffic_source'] = tr, src
        if _hip.WINOGRAD and args.model == 'darknet' and not args.no_direct_leg:
            _hip.WINOGRAD = False
            dnn._plan_cache = None
            try:
                ddt, _, dtable, _ = measure(min(args.steps, 20), 2, True)
                droof = roofline_from(dtable, 'the same step with every 3x3 layer on the implicit-GEMM kernel (Winograd off): executed == algorithmic FLOPs')
                roof['direct_only'] = {'images_per_sec': round(args.batch * min(args.steps, 20) / ddt, 2), 'kernel': droof['kernel'], 'achieved': droof['achieved'],
                                       'frac': droof['frac'], 'all_mfma_kernels': droof['all_mfma_kernels']}
            except Exception as e:
                roof['direct_only'] = {'error': '%s: %s' % (type(e).__name__, e)}
            finally:
                _hip.WINOGRAD = True
                dnn._plan_cache = None
        if _hip.WINOGRAD and not _hip.SPLIT and args.model == 'darknet' and not args.no_split_leg:
            # opt-in precision modes, reported BESIDE the fp32-MFMA headline: the Winograd GEMMs of the layers where it measures faster run on the
            # bf16 / fp16 matrix pipe from split operands (bf16x6: three bf16 planes, six plane products; f16x3: two scaled fp16 planes, three
            # products); same parity tests (tests/test_gpu_split.py, every test of tests/test_gpu_fullsize.py in each mode)
            with torch.no_grad():
                ref_feat = dnn.forward_nhwc(xs[0]).clone()
            for mode, tag, peak in (('bf16', 'split_bf16x6', PEAK_BF16_MFMA_TFLOPS / 6.0), ('f16', 'split_f16x3', PEAK_BF16_MFMA_TFLOPS / 3.0)):
                _hip.SPLIT = mode
                dnn._cache = None
                dnn._plan_cache = None
                try:
                    sdt, _, stable, _ = measure(min(args.steps, 20), 3, True)
                    with torch.no_grad():
                        got = dnn.forward_nhwc(xs[0])
                    plan = dnn._plan_cache[1]
                    sroof = roofline_from(stable, 'the same step in the %s mode' % tag)
                    gs = [r for r in sroof['top_kernels'] if r['kernel'].startswith('gemm_split')]
                    roof[tag] = {'dtype': ('f32 operands as 3 bf16 planes, 6 plane products per multiply on the bf16 MFMA pipe' if mode == 'bf16' else
                                           'f32 operands as 2 fp16 planes (fixed power-of-two scales), 3 plane products per multiply on the fp16 MFMA pipe') + ' (fp32 accumulate); transforms fp32',
                                 'images_per_sec': round(args.batch * min(args.steps, 20) / sdt, 2), 'ms_per_step': round(sdt / min(args.steps, 20) * 1e3, 4),
                                 'layers_on_split_gemm': int(sum(1 for i in range(plan['n']) if plan['arr'][i].algo in (4, 5))),
                                 'feature_max_abs_diff_over_rms_vs_fp32_mfma_plan': float(((got - ref_feat).abs().max() / ref_feat.pow(2).mean().sqrt()).item()),
                                 'gemm_split_kernel': gs[0] if gs else None, 'gemm_peak_tflops': round(peak, 1), 'kernel_ms_per_step': sroof['kernel_ms_per_step'],
                                 'parity': 'same tests as the fp32 path: tests/test_gpu_split.py (GEMM / conv vs fp64), tests/test_gpu_fullsize.py in this mode (2e-5 x rms vs fp64 at batch 32, 1e-4 IoU, bit-exact NMS on identical inputs)'}
                    if gs:
                        roof[tag]['gemm_split_kernel'] = dict(gs[0], frac=round(gs[0]['executed_tflops'] / peak, 4))
                    if mode == 'f16':
                        roof[tag]['operand_left_fp16_range'] = bool(_hip.split_overflowed())
                except Exception as e:
                    roof[tag] = {'error': '%s: %s' % (type(e).__name__, e)}
                finally:
                    _hip.SPLIT = ''
                    dnn._cache = None
                    dnn._plan_cache = None
    state = {k: v.detach().cpu() for k, v in dnn.state_dict().items()} if (ctx.world == 1 and args.cpu_sample > 0 and args.model == 'darknet') else None
    del inf, dnn
    release_memory()
    return out, roof, state, anchors


def conv3x3_leg(args, ctx):
    """north_star target quantity: MFMA utilisation of the 3x3 convolutions at 416x416 batch 64 (17 layers, 28.21 GFLOP/img).
    Per-layer HIP event pairs around each y2_conv_fwd of the inference plan (layers1.0 through its own y2_conv0_fwd)."""
    import ctypes

    import torch

    import _hip
    import bench_data
    B, S = 64, args.size
    inf, anchors = bench_data.build_model(args.classes, ctx.dev, 'darknet')
    dnn = inf.dnn
    x = bench_data.images(B, S, seed=5).to(ctx.dev)
    L, st = _hip.lib(), _hip.stream()

    def run(reps=5):
        with torch.no_grad():
            for _ in range(2):
                dnn.forward_nhwc(x)
        plan = dnn._plan_cache[1]
        alg = exe = ms = 0.0
        layers = 0
        # layers1.0 (K = 27) is one y2_conv0_fwd launch: time it through the kernel hooks
        t0 = kernel_table(lambda i: dnn.forward_nhwc(x), 2)
        if 'conv0_kernel' in t0:
            ms += t0['conv0_kernel']['ms']
            alg += plan['flops0']
            exe += plan['flops0']
            layers += 1
        for i in range(plan['n']):
            p = plan['arr'][i]
            if p.ksize != 3:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            L.y2_conv_fwd(ctypes.byref(p), st)
            e0.record()
            for _ in range(reps):
                L.y2_conv_fwd(ctypes.byref(p), st)
            e1.record()
            e1.synchronize()
            ms += e0.elapsed_time(e1) / reps
            a = 2.0 * p.Cin * p.Cout * 9 * p.B * p.H * p.W
            alg += a
            exe += 2.0 * p.Cin * p.Cout * 16 * p.B * ((p.H + 1) // 2) * ((p.W + 1) // 2) if p.algo in (1, 2, 3, 4, 5) else a
            layers += 1
        return {'layers': layers, 'ms': round(ms, 4), 'executed_tflops': round(exe / ms / 1e9, 2), 'mfma_utilisation': round(exe / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                'direct_equiv_tflops': round(alg / ms / 1e9, 2), 'algorithmic_gflop': round(alg / 1e9, 1), 'executed_gflop': round(exe / 1e9, 1)}
    out = {'workload': 'the 3x3 convolutions of Darknet-19 at %dx%d batch %d (inference plan, per-layer event pairs; transforms and fix-up kernels of a layer included in its time)' % (S, S, B),
           'target': 'north_star: >= 0.40 MFMA utilisation', 'peak': PEAK_FP32_MFMA_TFLOPS}
    out['autotuned'] = run()
    if _hip.WINOGRAD and not args.no_direct_leg:
        _hip.WINOGRAD = False
        dnn._plan_cache = None
        try:
            out['direct_only'] = run()
        finally:
            _hip.WINOGRAD = True
            dnn._plan_cache = None
    del inf, dnn
    release_memory()
    return out


def contended_step_ms(ctx, step, k=8, reps=6, hold_ms=60.0):
    """Rehearsal of the data-parallel step on ONE GPU (tools/contention.py): RCCL's all-reduce kernels hold CUs while backward
    runs and the persistent fused Winograd kernels want whole CUs.  A background stream holds k workgroup slots (256 threads + 32 KB LDS
    each, tools/probes/spin.hip) for the duration of a step; returns the median step time under that load beside the same measurement
    with nothing held.  A PREDICTION for the first 8-GPU run to be checked against, not a measurement of it."""
    import ctypes
    path = os.path.join(ROOT, 'tools', 'probes', 'libspin.so')
    if not os.path.exists(path):
        return None
    try:
        import torch
        spin = ctypes.CDLL(path)
        spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
        side = torch.cuda.Stream()
        res = {}
        for held in (0, k):
            ts = []
            for i in range(reps):
                ctx.sync()
                if held and spin.spin_launch(held, 256, hold_ms, ctypes.c_void_p(side.cuda_stream)) != 0:
                    return None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                step(i)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ctx.sync()
            res[held] = sorted(ts)[len(ts) // 2]
        return {'ms_per_step_contended': round(res[k], 3), 'ms_per_step_uncontended_same_protocol': round(res[0], 3),
                'contended_note': 'single GPU rehearsal: %d workgroup slots (256 threads, 32 KB LDS) held by a background kernel during the step (what RCCL channels look like to the '
                                  'persistent kernels); a prediction for the data-parallel step, unmeasured on 8-GPU hardware' % k}
    except Exception as e:
        print('contention rehearsal skipped (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
        return None


# ---------------------------------------------------------------------------------------------------- train leg
def train_leg(args, ctx):
    import torch

    import bench_data
    import train as y2train
    import utils
    steps = args.train_steps or args.steps          # EXACTLY K timed steps, like every leg (this is the headline leg at every N)
    B, S = args.train_batch, args.size
    nbatch = max(1, min(args.rotate, 2))
    data = []
    for i in range(nbatch):
        d = {k: v.to(ctx.dev) for k, v in bench_data.labels(B, S, args.classes, seed=2 + ctx.rank * 16 + i).items()}
        d['tensor'] = bench_data.images(B, S, seed=11 + ctx.rank * 16 + i).to(ctx.dev)
        data.append(d)

    def make(wrap):
        inf, anchors = bench_data.build_model(args.classes, ctx.dev, args.model)
        inf.train()
        m = y2train.ensure_model(inf) if wrap else inf
        if m is not inf:
            m.overlap_stats = []          # event pair per step around the wait for the outstanding all-reduces (dp_exposed_comm_ms_per_step)
        opt = utils.optim.SGD(m.parameters(), 1e-3, momentum=0.9)      # fused multi-tensor step (y2_opt_sgd), torch.optim.SGD semantics
        last = {}

        def step(i):
            last['r'] = y2train.iterate(m, opt, data[i % nbatch], bench_data.HPARAM, bench_data.THRESHOLD, anchors)
        return step, last, (inf, m, opt)

    per_img = FLOPS_TRAIN_PER_IMG * (S / 416.0) ** 2 if args.model == 'darknet' else None
    out = {'per_gpu_batch': B, 'global_batch': B * ctx.world, 'steps': steps, 'resident_batches_rotated': nbatch,
           'optimizer': 'utils.optim.SGD(lr=1e-3, momentum=0.9): fused multi-tensor HIP kernel, torch.optim.SGD semantics',
           'parallelism': 'dp%d: one process per GPU, bucketed RCCL all-reduce from inside backward + positive-count all-reduce' % ctx.world if ctx.world > 1 else 'single GPU'}
    single = None
    if ctx.world > 1:
        # the same step without the wrapper, every rank at once: per-GPU rate with zero communication (DP efficiency denominator)
        step, last, keep = make(False)
        for i in range(3):
            step(i)
        dt, _ = ctx.timed(step, max(4, steps // 2))
        single = B * max(4, steps // 2) / dt
        out['single_gpu_images_per_sec'] = round(single, 2)
        out['single_gpu_note'] = 'same step, no DP wrapper, all ranks running concurrently, MAX over ranks: per-GPU rate with zero communication'
        del step, last, keep
        release_memory()
    step, last, keep = make(True)
    for i in range(5):          # untimed: autotune, allocator, and (N > 1) the wrapper's one-time adoption of rank 0's algorithm choices at its 4th call
        step(i)
    dt, host = ctx.timed(step, steps)
    if ctx.world > 1:
        out['autotune_choices_synced'] = getattr(keep[1], 'tune_synced', None)
        st = getattr(keep[1], 'overlap_stats', None)
        if st:
            ctx.sync()
            ex = sorted(a.elapsed_time(b) for a, b in st[-steps:])
            out['dp_exposed_comm_ms_per_step'] = round(ex[len(ex) // 2], 3)
            out['dp_exposed_comm_note'] = 'median time the compute stream waits for the outstanding all-reduces after backward has finished (event pair around the waits)'
    out.update({'images_per_sec': round(B * steps * ctx.world / dt, 2), 'ms_per_step': round(dt / steps * 1e3, 3), 'host_ms_per_step': round(host / steps * 1e3, 3),
                'loss_total': float(last['r']['loss_total'].detach())})
    out['host_issue_ms_per_step'] = issue_time(ctx, step)
    if ctx.world == 1:
        c = contended_step_ms(ctx, step)
        if c is not None:
            out.update(c)
    runner = keep[0].__dict__.get('_y2_step_runner')
    out['launch'] = ('hipGraph replay of the captured step (%d graph segment(s)) + eager optimizer' % sum(1 for p in runner.plans.values() for op in (p.ops or []) if op[0] == 'graph')
                     if (runner is not None and runner.captures) else 'eager launches')
    if runner is not None and runner.plans:
        pl = next(iter(runner.plans.values()))
        out['operand_forms_prepared'] = None if pl.only is None else len(pl.only)      # None: the captured step derives every operand form (not the pruned step)
    if single:
        out['dp_speedup_vs_single_gpu'] = round(out['images_per_sec'] / single, 3)
    if per_img is not None:
        out['direct_equiv_tflops_per_gpu'] = round(per_img * B * steps / dt / 1e12, 2)
    if ctx.world == 1:
        # per-kernel durations are taken with the weight gradients on the MAIN stream: in the timed two-stream schedule they co-run with
        # the BatchNorm / transform passes of the next layers and the kernels inflate each other's durations (sum 60 ms in a 40 ms step)
        from model import train_graph
        streams, train_graph.BWD_STREAMS = train_graph.BWD_STREAMS, 1
        graph, y2train.GRAPH = y2train.GRAPH, False          # the event hooks bracket LAUNCHES: this table comes from the same launch sequence issued eagerly
        try:
            step(0)
            table = kernel_table(step, 2)
        finally:
            train_graph.BWD_STREAMS = streams
            y2train.GRAPH = graph
        std = B == 64 and S == 416 and args.model == 'darknet' and args.classes == 20
        out['roofline'] = r = roofline_from(table, 'training step, batch %d: fprop / dgrad / wgrad kernels with their executed FLOPs (the timed step = one hipGraph replay, weight gradients '
                                                   'forked onto a side stream, + the fused optimizer)' % B, trace_tag='train_b64' if std else None)
        r['kernel_ms_sum_single_stream'] = round(sum(e['ms'] for e in table.values()), 3)
        # the whole TIMED step against the fp32-MFMA peak: executed multiply-adds of all its MFMA kernels, and SURVEY.md 8d's algorithmic FLOPs (87.78 GFLOP per image)
        r['step_frac_executed'] = round(r['all_mfma_kernels']['executed_flops_per_step'] / (dt / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        if per_img is not None:
            r['step_frac_direct_equiv'] = round(per_img * B / (dt / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        r['traffic'], r['traffic_source'] = static_traffic('train_b64') if std else (None, None)
    del step, last, keep
    release_memory()
    return out


# ---------------------------------------------------------------------------------------------------- multi-scale train leg
def multiscale_leg(args, ctx):
    """BASELINE configs[3]: COCO-80 Darknet-19, batch 64 per GPU, the input size changes every `maintain` batches over 320..608 step 32
    (config.ini:39-40, utils/data.py:135-141: the collate function resizes the whole batch, so every rank switches at the same step).
    Protocol: one untimed pass over the schedule (first visit of every size: algorithm selection per new problem shape, allocator growth,
    under N > 1 the adoption of rank 0's algorithm table per new shape) whose per-size cost is REPORTED as first_visit_ms; then the timed
    region = `cycles` passes over the schedule (barrier + sync | sizes x maintain steps | barrier + sync, MAX over ranks); then one more
    pass with a synchronisation after every step for the per-size table and the cost of a size switch (first step at a size minus the
    median of the other steps at that size)."""
    import torch

    import bench_data
    import train as y2train
    import utils
    sizes = [int(v) for v in args.ms_sizes.split(',') if v]
    B, C, maintain = args.train_batch, args.ms_classes, max(2, args.ms_maintain)
    data = {}
    for S in sizes:
        d = {k: v.to(ctx.dev) for k, v in bench_data.labels(B, S, C, seed=2 + ctx.rank * 16 + S).items()}
        d['tensor'] = bench_data.images(B, S, seed=11 + ctx.rank * 16 + S).to(ctx.dev)
        data[S] = d
    inf, anchors = bench_data.build_model(C, ctx.dev, args.model)
    inf.train()
    m = y2train.ensure_model(inf) if ctx.world > 1 else inf
    opt = utils.optim.SGD(m.parameters(), 1e-3, momentum=0.9)
    last = {}

    def step(S):
        last['r'] = y2train.iterate(m, opt, data[S], bench_data.HPARAM, bench_data.THRESHOLD, anchors)

    import _hip
    first_visit, first_miss = {}, {}
    reserved0 = torch.cuda.memory_reserved() / 2.0 ** 30 if ctx.gpu else None
    # the sizes are known up front (config.ini:39 `[data] sizes`): the training steps' activation arena is sized ONCE for the largest (nothing executes:
    # train.reserve captures that size's step into the arena and drops it); reported, not part of any first visit
    reserve_ms = None
    if ctx.gpu and y2train.ARENA:
        ctx.sync()
        t0 = time.perf_counter()
        y2train.reserve(m, data[max(sizes)], bench_data.HPARAM, bench_data.THRESHOLD, anchors)
        ctx.sync()
        reserve_ms = round((time.perf_counter() - t0) * 1e3, 1)
    reserved1 = torch.cuda.memory_reserved() / 2.0 ** 30 if ctx.gpu else None
    for S in sizes:                      # untimed: first visit of every size
        ctx.sync()
        miss0 = len(_hip.TUNE_MISSES)
        t0 = time.perf_counter()
        for _ in range(max(5, maintain // 2)):
            step(S)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            step(S)
        ctx.sync()
        steady = (time.perf_counter() - t1) / 3
        first_visit[S] = max(0.0, (t1 - t0) - max(5, maintain // 2) * steady)
        first_miss[S] = len(_hip.TUNE_MISSES) - miss0          # problem shapes that had to be MEASURED at this visit (0 with a complete default table)
    schedule = [S for _ in range(max(1, args.ms_cycles)) for S in sizes for _ in range(maintain)]
    dt, host = ctx.timed(lambda i: step(schedule[i]), len(schedule))
    issue = {S: issue_time(ctx, lambda i, S=S: step(S), 4) for S in (sizes[0], sizes[-1])}
    images = B * len(schedule) * ctx.world
    # per-step table (synchronised after every step: NOT the throughput number)
    per = {S: [] for S in sizes}
    for S in sizes:
        for _ in range(maintain):
            ctx.sync()
            t0 = time.perf_counter()
            step(S)
            ctx.sync()
            per[S].append((time.perf_counter() - t0) * 1e3)
    table, switch = [], []
    for S in sizes:
        rest = sorted(per[S][1:])
        med = rest[len(rest) // 2]
        switch.append(max(0.0, per[S][0] - med))
        table.append({'size': S, 'ms_per_step': round(med, 3), 'images_per_sec': round(B * ctx.world / med * 1e3, 1), 'first_step_after_switch_ms': round(per[S][0], 3),
                      'switch_cost_ms': round(max(0.0, per[S][0] - med), 3), 'first_visit_ms': round(first_visit[S] * 1e3, 1), 'first_visit_shapes_measured': first_miss[S]})
    per_img = FLOPS_TRAIN_PER_IMG * sum((S / 416.0) ** 2 for S in sizes) / len(sizes) if args.model == 'darknet' else None
    out = {'workload': '%s YOLOv2 %d-class multi-scale train, batch-%d/GPU, sizes %s, resize every %d batches: fwd + region loss + bwd + SGD (BASELINE configs[3] per GPU)'
                       % (args.model, C, B, '..'.join(str(v) for v in (sizes[0], sizes[-1])) + ' step %d' % (sizes[1] - sizes[0] if len(sizes) > 1 else 0), maintain),
           'images_per_sec': round(images / dt, 2), 'steps': len(schedule), 'ms_per_step_mean': round(dt / len(schedule) * 1e3, 3), 'host_ms_per_step_mean': round(host / len(schedule) * 1e3, 3),
           'host_issue_ms_per_step': {str(S): v for S, v in issue.items()},
           'switch_cost_ms_mean': round(sum(switch) / len(switch), 3), 'switch_cost_ms_max': round(max(switch), 3),
           'first_visit_ms_mean': round(sum(first_visit.values()) / len(first_visit) * 1e3, 1), 'first_visit_ms_max': round(max(first_visit.values()) * 1e3, 1), 'first_visit_ms_total': round(sum(first_visit.values()) * 1e3, 1),
           'per_gpu_batch': B, 'global_batch': B * ctx.world, 'loss_total': float(last['r']['loss_total'].detach()),
           'first_visit_shapes_measured': sum(first_miss.values()), 'first_visit_measured_keys': [list(map(str, k)) for k in _hip.TUNE_MISSES[-8:]] if sum(first_miss.values()) else [],
           'default_tune_table_entries': _hip._DEFAULTS_SEEN.get(str(ctx.dev)),
           'reserved_gib_before_after': [None if reserved0 is None else round(reserved0, 1), round(torch.cuda.memory_reserved() / 2.0 ** 30, 1) if ctx.gpu else None],
           'arena_reserve_ms': reserve_ms, 'reserved_gib_after_reserve': None if reserved1 is None else round(reserved1, 1),
           'parallelism': 'dp%d' % ctx.world if ctx.world > 1 else 'single GPU', 'per_size': table}
    if per_img is not None:
        out['direct_equiv_tflops_per_gpu'] = round(per_img * B * len(schedule) / dt / 1e12, 2)
    if ctx.world > 1:
        out['autotune_choices_synced'] = getattr(m, 'tune_synced', None)
    del m, inf, opt, data
    torch.cuda.empty_cache()
    return out



# ---------------------------------------------------------------------------------------------------- single-image / small-batch latency leg
MIN_BYTES_B1 = 303e6       # every conv reads input + weights and writes its output once, fp32, 416x416, batch 1 (202.6 MB of it weights)
WEIGHT_BYTES = 202.6e6
PEAK_HBM_TBS = 8.0


def latency_leg(args, ctx):
    """BASELINE configs[0] is the reference's own usage: detect.py feeds ONE image per call (detect.py:141-153).  Its GPU twin: the
    same conv + decode + filter + NMS step at batch 1 and batch 8, one captured hipGraph, strictly serial replays (latency, not
    throughput).  Floors beside it: algorithmic conv FLOPs / fp32-MFMA peak and one-pass HBM bytes / 8 TB/s - at batch 1 both are
    far below the measured time: the step is a chain of ~30 dependent launches on maps as small as 13x13."""
    import torch

    import bench_data
    import detect
    inf, anchors = bench_data.build_model(args.classes, ctx.dev, 'darknet')
    dnn = inf.dnn
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
    out = {'workload': 'Darknet-19 YOLOv2 %dx%d single-image / batch-8 inference: conv stack + decode + filter + NMS, hipGraph, serial (detect.py:141-153 on the GPU)' % (args.size, args.size)}
    for B in (1, 8):
        x = bench_data.images(B, args.size, seed=40 + B).to(ctx.dev)

        def eager(i):
            with torch.no_grad():
                return detect.detect_batch(dnn.forward_nhwc(x), anchors, **kw)
        for i in range(3):
            eager(i)
        ctx.sync()
        table = kernel_table(eager, 4)
        plan = dnn._plan_cache[1]
        names = ['layers1.0'] + [n for n, _, _ in sum(dnn._blocks(), [])][1:]
        algo_names = {0: 'direct', 1: 'winograd', 2: 'wino-fused', 3: 'wino-implicit', 4: 'split-bf16', 5: 'split-f16', 6: 'wino-f43'}
        blk_name = {m: n for n, m, _ in sum(dnn._blocks(), [])}
        blk_name[dnn.passthrough] = 'passthrough'
        layers = [{'layer': blk_name.get(b, '?'), 'algo': algo_names.get(int(p.algo), str(p.algo)), 'tile': int(p.tile), 'HxW': '%dx%d' % (p.H, p.W), 'Cin': int(p.Cin), 'Cout': int(p.Cout), 'k': int(p.ksize)}
                  for p, b in zip(plan['arr'], plan['blks'])]
        g = detect.GraphedDetector(dnn, anchors, x, static_input=True, **kw)
        for _ in range(20):
            g.run()
        steps = 200
        dt, host = ctx.timed(lambda i: g.run(), steps)
        ms = dt / steps * 1e3
        flops = FLOPS_FWD_PER_IMG * B * (args.size / 416.0) ** 2
        hbm = (WEIGHT_BYTES + (MIN_BYTES_B1 - WEIGHT_BYTES) * B) * (1.0 if args.size == 416 else (args.size / 416.0) ** 2)
        mfma_floor = flops / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
        hbm_floor = hbm / (PEAK_HBM_TBS * 1e12) * 1e3
        kms = sum(e['ms'] for e in table.values())
        executed = float(plan['flops_executed'] + plan['flops0']) if 'flops_executed' in plan else flops      # Winograd layers execute 16/36 of their direct count
        out['b%d' % B] = {'batch': B, 'ms_per_step': round(ms, 4), 'ms_per_image': round(ms / B, 4), 'images_per_sec': round(B / ms * 1e3, 1), 'host_ms_per_step': round(host / steps * 1e3, 4),
                          'launches_per_step': round(sum(e['launches'] for e in table.values()), 1), 'kernel_ms_sum_eager': round(kms, 4),
                          'algorithmic_gflop': round(flops / 1e9, 2), 'min_hbm_mbytes': round(hbm / 1e6, 1),
                          'mfma_floor_ms': round(mfma_floor, 4), 'hbm_floor_ms': round(hbm_floor, 4), 'bound': 'mfma' if mfma_floor >= hbm_floor else 'hbm',
                          'direct_equiv_frac': round(mfma_floor / ms, 4), 'direct_equiv_note': 'ALGORITHMIC conv FLOPs / fp32-MFMA peak / time: not a roofline fraction when layers run Winograd',
                          'executed_gflop': round(executed / 1e9, 2), 'executed_mfma_frac': round(executed / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                          'achieved_tflops_direct_equiv': round(flops / ms / 1e9, 2), 'achieved_hbm_tbs_min_bytes': round(hbm / ms / 1e9, 3),
                          'weight_bytes_tbs': round(WEIGHT_BYTES / ms / 1e9, 3),
                          'winograd_layers': int(sum(1 for l in layers if l['algo'] != 'direct')), 'plan': layers,
                          'top_kernels': top_kernels(table, 0.03)[0]}
        del g
    del inf, dnn
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------- ResNet-50 608x608 COCO-80 leg (configs[4])
RESNET50_608_FWD_GFLOP = 60.85          # SURVEY.md 8d [probe]: model/resnet.py resnet50 plugin, 608x608, 80 classes, per image
RESNET50_608_STEM_DGRAD_GFLOP = 2 * 3 * 64 * 49 * 304 * 304 / 1e9       # the data gradient of the 7x7 stem is never computed


def resnet_leg(args, ctx):
    """BASELINE configs[4] per GPU: the plugin swap (`[model] dnn = model.resnet.resnet50`, config.ini:25) at 608x608 with the COCO-80 head:
    batch-32 inference (conv stack + decode + filter + NMS, hipGraph) and batch-32 training (fwd + region loss + bwd + SGD, StepPlan),
    with the per-kernel table of the training step (launch sequence issued eagerly under the event hooks)."""
    import torch

    import bench_data
    import detect
    import train as y2train
    import utils
    S, C, B = 608, 80, args.resnet_batch
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
    out = {'workload': 'model.resnet.resnet50 YOLOv2 608x608 COCO-80, batch-%d/GPU (BASELINE configs[4] per GPU)' % B, 'per_gpu_batch': B}
    inf, anchors = bench_data.build_model(C, ctx.dev, 'resnet50')
    dnn = inf.dnn
    x = bench_data.images(B, S, seed=71 + ctx.rank).to(ctx.dev)
    with torch.no_grad():
        for _ in range(3):
            detect.detect_batch(dnn.forward_nhwc(x), anchors, **kw)
    ctx.sync()
    g = detect.GraphedDetector(dnn, anchors, x, static_input=True, **kw)
    for _ in range(3):
        g.run()
    steps = 10
    dt, _ = ctx.timed(lambda i: g.run(), steps)
    out['detect'] = {'images_per_sec': round(B * steps * ctx.world / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 3),
                     'direct_equiv_tflops_per_gpu': round(RESNET50_608_FWD_GFLOP * B * steps / dt / 1e3, 2)}
    out['detect']['direct_equiv_frac'] = round(out['detect']['direct_equiv_tflops_per_gpu'] / PEAK_FP32_MFMA_TFLOPS, 4)
    del g
    inf.train()
    m = y2train.ensure_model(inf) if ctx.world > 1 else inf
    opt = utils.optim.SGD(m.parameters(), 1e-3, momentum=0.9)
    d = {k: v.to(ctx.dev) for k, v in bench_data.labels(B, S, C, seed=72 + ctx.rank).items()}
    d['tensor'] = x
    last = {}

    def step(i):
        last['r'] = y2train.iterate(m, opt, d, bench_data.HPARAM, bench_data.THRESHOLD, anchors)
    for i in range(5):
        step(i)
    steps = 6
    dt, host = ctx.timed(step, steps)
    per_img = (3 * RESNET50_608_FWD_GFLOP - RESNET50_608_STEM_DGRAD_GFLOP) * 1e9
    out['train'] = {'images_per_sec': round(B * steps * ctx.world / dt, 1), 'ms_per_step': round(dt / steps * 1e3, 3), 'host_ms_per_step': round(host / steps * 1e3, 3),
                    'direct_equiv_tflops_per_gpu': round(per_img * B * steps / dt / 1e12, 2), 'loss_total': float(last['r']['loss_total'].detach())}
    out['train']['direct_equiv_frac'] = round(out['train']['direct_equiv_tflops_per_gpu'] / PEAK_FP32_MFMA_TFLOPS, 4)
    if ctx.world == 1:
        graph, y2train.GRAPH = y2train.GRAPH, False
        try:
            step(0)
            table = kernel_table(step, 2)
        finally:
            y2train.GRAPH = graph
        roof = roofline_from(table, 'ResNet-50 608x608 COCO-80 training step, batch %d (same launch sequence issued eagerly under the event hooks)' % B)
        out['train']['roofline'] = roof
    del m, inf, dnn, opt
    torch.cuda.empty_cache()
    return out

# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(sd, anchors, size, sample):
    """Oracle (port of the reference path) on the host cores: conv stack + decode + filter + NMS on `sample` images.  The ONLY
    place bench.py touches `oracle/`: a reported baseline beside the GPU number, never part of a GPU leg."""
    import torch
    from oracle import darknet as odark
    from oracle import detect as odet
    from oracle import head as ohead
    import bench_data
    cores = os.cpu_count() or 1
    x = bench_data.images(min(sample, 16), size, seed=1).repeat((sample + 15) // 16, 1, 1, 1)[:sample]
    with torch.no_grad():
        best = (1e30, cores)          # the thread count that serves the oracle best on this host (oneDNN degrades when oversubscribed)
        for th in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8)}):
            torch.set_num_threads(th)
            odark.forward(x[:2], sd)
            t0 = time.perf_counter()
            odark.forward(x[:2], sd)
            best = min(best, (time.perf_counter() - t0, th))
        torch.set_num_threads(best[1])
        t0 = time.perf_counter()
        for i0 in range(0, sample, 16):
            feat = odark.forward(x[i0:i0 + 16], sd)
            pred = ohead.decode(feat, anchors)
            B = feat.shape[0]
            prob = torch.softmax(pred['logits'], -1).view(B, -1, pred['logits'].shape[-1]).numpy()
            iou = pred['iou'].reshape(B, -1).numpy()
            mn, mx = pred['yx_min'].reshape(B, -1, 2).numpy(), pred['yx_max'].reshape(B, -1, 2).numpy()
            for b in range(B):
                odet.postprocess(iou[b], mn[b], mx[b], prob[b], fix=True)
        dt = time.perf_counter() - t0

        def one_image(img):
            feat = odark.forward(img, sd)
            pred = ohead.decode(feat, anchors)
            prob = torch.softmax(pred['logits'], -1).view(1, -1, pred['logits'].shape[-1]).numpy()
            return odet.postprocess(pred['iou'].reshape(-1).numpy(), pred['yx_min'].reshape(-1, 2).numpy(), pred['yx_max'].reshape(-1, 2).numpy(), prob[0], fix=True)
        # ---- BASELINE configs[0]: the reference's own usage, ONE image per call (detect.py:141-153)
        one_image(x[:1])
        n1 = 8
        t0 = time.perf_counter()
        for i in range(n1):
            one_image(x[i:i + 1])
        b1_ms = (time.perf_counter() - t0) / n1 * 1e3
    # ---- one batch-8 training step (train.py:338-362 semantics: fwd with batch statistics + region loss + backward + SGD), SURVEY.md 8d
    from oracle import loss as oloss
    Bt = 8
    sdt = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
    opt = torch.optim.SGD([v for v in sdt.values() if v.requires_grad], 1e-3, momentum=0.9)
    xt = bench_data.images(Bt, size, seed=11)
    lab = bench_data.labels(Bt, size, 20, seed=2)
    rows = size // 32
    scale = torch.tensor([rows / size, rows / size]).view(1, 1, 2)
    data = dict(yx_min=lab['yx_min'] * scale, yx_max=lab['yx_max'] * scale, cls=lab['cls'])
    tt = []
    for i in range(3):
        t0 = time.perf_counter()
        lo, _ = oloss.loss(anchors, data, ohead.decode(odark.forward(xt, sdt, training=True), anchors), bench_data.THRESHOLD)
        opt.zero_grad()
        oloss.total(lo, bench_data.HPARAM).backward()
        opt.step()
        tt.append(time.perf_counter() - t0)
    train_s = min(tt[1:])
    # ---- greedy NMS at n = 200 candidates (utils/postprocess.py:23-49), the reference's per-image post-processing cost
    from oracle import nms as onms
    from oracle import synth
    sc, mn, mx = synth.nms_boxes(200)
    onms.nms(sc, mn, mx, 0.45, 200)
    t0 = time.perf_counter()
    for _ in range(20):
        onms.nms(sc, mn, mx, 0.45, 200)
    nms_ms = (time.perf_counter() - t0) / 20 * 1e3
    cpu_model = None
    try:
        cpu_model = next(l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except Exception:
        pass
    return {'value': round(sample / dt, 3), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port', 'cpu_model': cpu_model, 'cpu_count': cores,
            'sample': '%d synthetic %dx%d images in batches of 16, oracle conv stack (torch-CPU fp32, best of %d/%d/%d/%d threads) + decode + filter(fix=1) + NMS, %.1f s'
                      % (sample, size, size, cores, cores // 2, cores // 4, cores // 8, dt),
            'b1_ms_per_image': round(b1_ms, 2), 'b1_images_per_sec': round(1e3 / b1_ms, 2), 'b1_sample': '%d calls of ONE %dx%d image: conv stack + decode + filter + NMS (BASELINE configs[0], detect.py:141-153)' % (n1, size, size),
            'train_b8_images_per_sec': round(Bt / train_s, 3), 'train_b8_s_per_step': round(train_s, 3),
            'train_b8_sample': 'best of 2 batch-8 %dx%d VOC-20 steps after one warm-up: oracle fwd (batch-stat BN) + region loss + torch-CPU autograd + SGD' % (size, size),
            'nms_n200_ms': round(nms_ms, 3), 'nms_sample': 'oracle.nms (numpy restatement of utils/postprocess.py:23-49), 200 candidates, overlap 0.45, mean of 20 calls'}


# ---------------------------------------------------------------------------------------------------- dry run (no GPU)
def dry_run(args, ctx):
    """Stand-in CPU workload so that launch, rendezvous, the DP wrapper and the timing protocol can be exercised without a GPU.
    Nothing of the hot path runs here (it has no CPU fallback): the printed numbers are meaningless and flagged as such."""
    import torch
    import torch.nn as nn

    import train as y2train
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(64, 256), nn.Tanh(), nn.Linear(256, 64))
    m = y2train.ensure_model(net)
    opt = torch.optim.SGD(m.parameters(), 1e-3, momentum=0.9)
    x = torch.randn(args.train_batch, 64)

    def train_step(i):
        opt.zero_grad()
        m(x).pow(2).mean().backward()
        opt.step()

    def detect_step(i):
        with torch.no_grad():
            net(x)
    for i in range(args.warmup):
        train_step(i)
    dt_t, _ = ctx.timed(train_step, args.steps)
    dt_d, _ = ctx.timed(detect_step, args.steps)
    wrapped = type(m).__name__
    return {'images_per_sec': round(args.train_batch * args.steps * ctx.world / dt_t, 2), 'ms_per_step': round(dt_t / args.steps * 1e3, 4), 'wrapper': wrapped}, \
           {'images_per_sec': round(args.train_batch * args.steps * ctx.world / dt_d, 2), 'ms_per_step': round(dt_d / args.steps * 1e3, 4)}



LINE_LIMIT = 6000        # bytes of the final stdout line: the driver parses the LAST line of stdout and keeps a bounded tail of it (round 4's 30 KB line was not parsed)
ROOF_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_launch_us', 'frac_uncontended', 'step_frac_executed', 'step_frac_direct_equiv',
             'kernel_share_of_step', 'launches_per_step', 'kernel_ms_per_step', 'frac_source', 'what', 'traffic_source')
# scalars that go first when the line would exceed LINE_LIMIT (least important first)
DROP_ORDER = ('split_bf16x6_detect_images_per_sec', 'split_f16x3_detect_images_per_sec', 'latency_b8_launches', 'latency_b1_launches', 'resnet50_608_train_kernel_ms_sum',
              'resnet50_608_train_mfma_ms_per_step', 'train_mfma_ms_per_step', 'train_kernel_ms_sum_single_stream', 'train_dominant_avg_launch_us', 'multiscale_switch_cost_ms_max',
              'latency_b8_direct_equiv_frac', 'latency_b1_direct_equiv_frac', 'detect_streams', 'resnet50_608_detect_images_per_sec', 'multiscale_ms_per_step_mean')


def scalar_name(kernel):
    return kernel.replace('[', '_').replace(']', '').replace('<', '_').replace('>', '').replace(',', '_').replace(' ', '')


def write_tables(full, path):
    """The long form of the result (per-kernel tables, plans, per-size tables): a JSON file next to the run, never the stdout line."""
    path = path or os.path.join(ROOT, 'gpurun_out', 'bench_full.json')
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(full, f)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except Exception as e:
        print('bench.py: tables not written (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
        return None


def compact_line(head, roof, extra, cb, tables_at=None, limit=LINE_LIMIT):
    """The ONE stdout line: top-level contract keys + config + roofline (scalars only, each once) + cpu_baseline, at most `limit` bytes."""
    out = dict(head)
    if roof is not None:
        r = {k: roof[k] for k in ROOF_KEYS if k in roof and not isinstance(roof[k], (dict, list))}
        for k in ('what', 'traffic_source', 'frac_source'):
            if isinstance(r.get(k), str) and len(r[k]) > 160:
                r[k] = r[k][:157] + '...'
        r.update({k: v for k, v in extra.items() if not isinstance(v, (dict, list))})
        out['roofline'] = r
    else:
        # N > 1: no per-kernel table is taken (the roofline and the CPU baseline are the N = 1 run's); the legs' scalars still belong in the line
        out['summary'] = {k: v for k, v in extra.items() if not isinstance(v, (dict, list))}
    if cb is not None:
        keep = ('value', 'unit', 'cores', 'kind', 'sample', 'cpu_model', 'cpu_count', 'b1_ms_per_image', 'train_b8_images_per_sec', 'nms_n200_ms', 'error')
        c = {k: cb[k] for k in keep if k in cb}
        if isinstance(c.get('sample'), str) and len(c['sample']) > 200:
            c['sample'] = c['sample'][:197] + '...'
        out['cpu_baseline'] = c
    if tables_at:
        out['tables'] = tables_at
    line = json.dumps(out, separators=(',', ':'))
    drop = list(DROP_ORDER)
    box = out.get('roofline', out.get('summary'))
    while len(line) > limit and box:
        k = drop.pop(0) if drop else next((k for k in reversed(list(box)) if k not in ROOF_KEYS[:11]), None)
        if k is None:
            break
        box.pop(k, None)
        line = json.dumps(out, separators=(',', ':'))
    return line


def naming(args, headline):
    """(metric, workload of the detect leg, workload of the train leg) - functions of the arguments only, never of the world size: the line of an
    N-GPU run names the same workload as the line of the 1-GPU run it is compared with."""
    label = {'darknet': 'Darknet-19', 'tiny': 'tiny-yolo'}.get(args.model, args.model)
    ref = {'darknet': (' (BASELINE configs[1])', ' (BASELINE configs[2])')}.get(args.model, (' (plugin swap: forward of BASELINE configs[4])', ' (plugin swap, BASELINE configs[4] per GPU)') if args.model.startswith('resnet') else ('', ''))
    det_workload = '%s YOLOv2 %dx%d batch-%d/GPU inference: conv stack + decode + filter + NMS%s' % (label, args.size, args.size, args.batch, ref[0])
    tr_workload = '%s YOLOv2 %d-class train %dx%d batch-%d/GPU: fwd + region loss + bwd + SGD%s' % (label, args.classes, args.size, args.size, args.train_batch, ref[1])
    if headline == 'train':
        metric = 'images/sec (%dx%d) train+detect, %s YOLOv2: value = TRAIN step, batch %d per GPU (configs[2]) at every N; detect (configs[1]) = roofline.detect_images_per_sec' % (args.size, args.size, label, args.train_batch)
    else:
        metric = 'images/sec (%dx%d) train+detect, %s YOLOv2: value = DETECT (configs[1]); train (configs[2]) = roofline.train_images_per_sec' % (args.size, args.size, label)
    return metric, det_workload, tr_workload


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    ctx = Ctx(args)
    torch = ctx.torch
    if ctx.world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or let bench.py launch the ranks itself)' % (args.gpus, ctx.world, args.gpus))
    if ctx.ranks_seen != ctx.world:
        raise SystemExit('bench.py: all-reduce of ones saw %d ranks, expected %d' % (ctx.ranks_seen, ctx.world))
    # ONE headline workload at every N (value(8) / value(1) must compare like with like): the batch-64 training step, the leg with a collective and the
    # configuration north_star's targets are quoted on; detect (configs[1], replicas) is reported beside it
    headline = args.headline if args.headline != 'auto' else 'train'
    if args.no_detect and headline == 'detect':
        headline = 'train'
    if args.no_train and headline == 'train':
        headline = 'detect'
    label = {'darknet': 'Darknet-19', 'tiny': 'tiny-yolo'}.get(args.model, args.model)

    if args.dry_run:
        tr, de = dry_run(args, ctx)
        if ctx.rank == 0:
            src = tr if headline == 'train' else de
            metric, det_workload, tr_workload = naming(args, headline)
            print(json.dumps({'metric': 'DRY RUN (stand-in CPU workload, launch/rendezvous/DP-wrapper/timing protocol only) of: ' + metric, 'value': src['images_per_sec'], 'unit': 'images/sec',
                              'n_gpus': ctx.world, 'ranks_seen': ctx.ranks_seen, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': src['ms_per_step'],
                              'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'dry_run': True, 'valid': False,
                              'headline': headline, 'config': {'workload': tr_workload if headline == 'train' else det_workload, 'executed': 'none of it: 2-layer CPU MLP stand-in',
                                                               'global_batch': (args.train_batch if headline == 'train' else args.batch) * ctx.world}, 'train': tr, 'detect': de}))
        if ctx.world > 1:
            ctx.dist.destroy_process_group()
        return
    assert ctx.gpu, 'bench.py needs an MI355X (use --dry-run to exercise the launch path without one)'

    det = roof = state = anchors = None
    if args.multiscale:
        args.no_detect = args.no_train = args.no_conv3 = args.no_latency = args.no_resnet = True
        args.cpu_sample = 0
    # The multi-scale leg runs FIRST, in a process that has not allocated anything yet - the state of a training job that meets a new size.  Its first-visit figures
    # are capture + growth of the graph pool by ~15 GB per size; what a hipMalloc of that size costs depends on the BOX (10-12 ms at every size in four consecutive
    # processes on one, 0.5-0.9 s from 480-544 upwards on others - whatever this bench did before the leg: DESIGN.md 3.7).
    ms = None
    if args.multiscale or (not args.no_multiscale and not args.no_train and args.model == 'darknet'):
        ctx.sync()
        time.sleep(args.settle)
        try:
            ms = multiscale_leg(args, ctx)
        except Exception as e:
            if args.multiscale:
                raise
            import traceback
            traceback.print_exc()
            ms = {'error': '%s: %s' % (type(e).__name__, e)}
    release_memory()
    if ms is not None and ctx.gpu and not args.multiscale:
        ctx.sync()
        time.sleep(args.settle)          # (a leg measures lower right behind seconds of full load: DESIGN.md 5)
    if not args.no_detect:
        det, roof, state, anchors = detect_leg(args, ctx)
    conv3 = None
    if ctx.world == 1 and args.model == 'darknet' and not args.no_conv3:
        try:
            conv3 = conv3x3_leg(args, ctx)
        except Exception as e:
            import traceback
            traceback.print_exc()
            conv3 = {'error': '%s: %s' % (type(e).__name__, e)}
    tr = None
    if not args.no_train:
        ctx.sync()
        time.sleep(args.settle)          # untimed pause between the inference legs and the training leg
        try:
            tr = train_leg(args, ctx)
        except Exception as e:
            if headline == 'train':
                raise
            import traceback
            traceback.print_exc()
            tr = {'error': '%s: %s' % (type(e).__name__, e)}
    lat = rn = None
    if ctx.world == 1 and args.model == 'darknet' and not args.no_latency and not args.no_detect:
        try:
            lat = latency_leg(args, ctx)
        except Exception as e:
            import traceback
            traceback.print_exc()
            lat = {'error': '%s: %s' % (type(e).__name__, e)}
    if args.model == 'darknet' and not args.no_resnet and not args.no_train:
        ctx.sync()
        try:
            rn = resnet_leg(args, ctx)
        except Exception as e:
            import traceback
            traceback.print_exc()
            rn = {'error': '%s: %s' % (type(e).__name__, e)}
    if ctx.rank == 0:
        metric, det_workload, tr_workload = naming(args, headline)
        if args.multiscale:
            value, msps, steps, workload = ms['images_per_sec'], ms['ms_per_step_mean'], ms['steps'], ms['workload']
            metric = 'images/sec (320..608 multi-scale) train, %s YOLOv2 (BASELINE configs[3] per GPU)' % label
            headline = 'multiscale'
        elif headline == 'train':
            value, msps, steps, workload = tr['images_per_sec'], tr['ms_per_step'], tr['steps'], tr_workload
        else:
            value, msps, steps, workload = det['images_per_sec'], det['ms_per_step'], det['steps'], det_workload
        ok = lambda d: d is not None and 'error' not in d
        # ---- scalars of the ONE line the driver parses (it keeps `roofline`, `config`, `cpu_baseline`): every leg's headline numbers, once
        extra = {}
        droof, troof = roof, (tr.get('roofline') if ok(tr) else None)
        if ok(det):
            extra.update(detect_images_per_sec=det['images_per_sec'], detect_ms_per_step=det['ms_per_step'], detect_streams=det.get('streams'),
                         detect_serial_images_per_sec=det.get('serial_images_per_sec'), detect_serial_ms_per_step=det.get('serial_ms_per_step'),
                         detect_to_host_images_per_sec=det.get('to_host_images_per_sec'))
        if droof is not None:
            if 'conv_chain' in droof:
                extra.update(conv_chain_ms_per_step=droof['conv_chain']['ms_per_step'], conv_chain_frac=droof['conv_chain']['frac'])
            if isinstance(droof.get('direct_only'), dict) and 'frac' in droof['direct_only']:
                extra['detect_direct_only_frac'] = droof['direct_only']['all_mfma_kernels']['frac']
            for r in droof.get('top_kernels', []):
                if r['frac'] is not None:       # one scalar per MFMA kernel of the detect step (uncontended single-stream launches): frac_<kernel>
                    extra['frac_' + scalar_name(r['kernel'])] = r['frac']
        if ok(tr):
            extra.update(train_images_per_sec=tr['images_per_sec'], train_ms_per_step=tr['ms_per_step'], train_host_issue_ms_per_step=tr.get('host_issue_ms_per_step'))
            for k in ('dp_exposed_comm_ms_per_step', 'ms_per_step_contended', 'single_gpu_images_per_sec'):
                if k in tr:
                    extra['train_' + k] = tr[k]
        # the leg that is NOT the headline hands its roofline scalars over with a prefix; the headline leg's ARE the `roofline` object
        side, prefix = (droof, 'detect_') if headline == 'train' else (troof, 'train_')
        if side:
            for k in ('kernel', 'frac', 'frac_uncontended', 'avg_launch_us', 'step_frac_executed', 'step_frac_direct_equiv', 'traffic', 'kernel_ms_per_step'):
                if side.get(k) is not None:
                    extra[prefix + ('dominant_' if k in ('kernel', 'frac', 'frac_uncontended', 'avg_launch_us') else '') + k] = side[k]
        if troof:
            extra.update(train_mfma_frac=troof['all_mfma_kernels']['frac'], train_mfma_ms_per_step=troof['all_mfma_kernels']['ms_per_step'],
                         train_kernel_ms_sum_single_stream=troof.get('kernel_ms_sum_single_stream'))
        if headline == 'train' and troof:
            roof = dict(troof)
        if ok(conv3):
            extra['conv3x3_b64_mfma_util'] = conv3['autotuned']['mfma_utilisation']
            if 'direct_only' in conv3:
                extra['conv3x3_b64_direct_only_util'] = conv3['direct_only']['mfma_utilisation']
        if ok(lat):
            for B in (1, 8):
                e = lat['b%d' % B]
                extra.update({'latency_b%d_ms' % B: e['ms_per_step'], 'latency_b%d_launches' % B: e['launches_per_step'],
                              'latency_b%d_executed_mfma_frac' % B: e['executed_mfma_frac'], 'latency_b%d_direct_equiv_frac' % B: e['direct_equiv_frac']})
            extra['latency_b1_weight_tbs'] = lat['b1']['weight_bytes_tbs']
        if ok(rn):
            extra.update(resnet50_608_detect_images_per_sec=rn['detect']['images_per_sec'], resnet50_608_train_images_per_sec=rn['train']['images_per_sec'],
                         resnet50_608_train_ms_per_step=rn['train']['ms_per_step'], resnet50_608_train_direct_equiv_frac=rn['train']['direct_equiv_frac'])
            r = rn['train'].get('roofline')
            if r:
                extra.update(resnet50_608_train_mfma_frac=r['all_mfma_kernels']['frac'], resnet50_608_train_mfma_ms_per_step=r['all_mfma_kernels']['ms_per_step'],
                             resnet50_608_train_kernel_ms_sum=r['kernel_ms_per_step'])
        if ok(ms):
            extra.update(multiscale_images_per_sec=ms['images_per_sec'], multiscale_ms_per_step_mean=ms['ms_per_step_mean'], multiscale_switch_cost_ms_max=ms['switch_cost_ms_max'],
                         multiscale_first_visit_ms_mean=ms['first_visit_ms_mean'], multiscale_first_visit_ms_max=ms.get('first_visit_ms_max'),
                         multiscale_first_visit_shapes_measured=ms.get('first_visit_shapes_measured'), multiscale_reserved_gib=(ms.get('reserved_gib_before_after') or [None, None])[1],
                         multiscale_arena_reserve_ms=ms.get('arena_reserve_ms'))
        if droof is not None:
            for tag in ('split_bf16x6', 'split_f16x3'):
                sp = droof.get(tag)
                if isinstance(sp, dict) and 'images_per_sec' in sp:
                    extra[tag + '_detect_images_per_sec'] = sp['images_per_sec']
        if roof is None and troof:
            roof = dict(troof)
        head = {'metric': metric, 'value': value, 'unit': 'images/sec', 'n_gpus': ctx.world, 'ranks_seen': ctx.ranks_seen, 'steps': steps, 'warmup': args.warmup,
                'ms_per_step': msps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'headline': headline,
                'config': {'workload': workload, 'classes': args.classes, 'global_batch': (args.batch if headline == 'detect' else args.train_batch) * ctx.world,
                           'parallelism': (ms if args.multiscale else tr if headline == 'train' else det)['parallelism'], 'weights': 'random-init seed 0 (bench_data.randomize)'}}
        cb = None
        if state is not None:
            try:
                cb = cpu_baseline(state, anchors, args.size, args.cpu_sample)
            except Exception as e:
                cb = {'error': '%s: %s' % (type(e).__name__, e)}
        # ---- everything (per-kernel tables, per-layer plans, per-size tables) goes to a FILE; stdout carries one compact line
        full = dict(head)
        for k, v in (('conv3x3_b64', conv3), ('detect', None if det is None else dict(det, workload=det_workload)), ('train', None if tr is None else dict(tr, workload=tr_workload)),
                     ('multiscale', ms), ('latency', lat), ('resnet50_608', rn), ('roofline', roof), ('cpu_baseline', cb)):
            if v is not None:
                full[k] = v
        full['scalars'] = extra
        tables_at = write_tables(full, args.tables)
        print(compact_line(head, roof, extra, cb, tables_at))
    if ctx.world > 1:
        ctx.dist.destroy_process_group()


if __name__ == '__main__':
    main()
