"""`train` — the data-parallel wrapper and step helpers of the reference's train.py, MI355X-native.

Reference: `ensure_model` (train.py:65-71) wraps the Inference module in single-process `nn.DataParallel`: every step
it re-broadcasts 202.6 MB of parameters from GPU0, scatters the batch, gathers outputs to GPU0 and reduce-adds all
gradients into GPU0.  Here: one process per GPU (launched by `python -m torch.distributed.run`), each rank owns a full
replica and a per-GPU batch (`-b` per GPU, global batch = b x N exactly like train.py:309), gradients are averaged with
bucketed RCCL all-reduces over xGMI that start INSIDE backward as soon as a bucket's layers have produced their weight
gradients (the Darknet backward calls `grad_ready_hook` per layer), BatchNorm statistics stay per replica (as in
nn.DataParallel) and rank 0's running statistics are the ones a checkpoint sees.  No parameter broadcast per step, no
output gather.  The cls term's mean over positives uses the GLOBAL positive count (one scalar all-reduce) so that the
averaged gradient equals the single-process gradient on the concatenated batch.

Only the hot-path pieces of train.py are mirrored (`norm_data` :57-62, `ensure_model` :65-71, the body of
`Train.iterate` :338-362 as `iterate`); the dataset / TensorBoard / checkpoint harness is out of scope.
"""
import logging
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

import model
import utils

BUCKET_BYTES = 25 * 1024 * 1024


_SCALES = {}


def norm_data(data, height, width, rows, cols, keys='yx_min, yx_max'):
    """train.py:57-62: ground-truth boxes from pixels to grid-cell units."""
    _data = {key: data[key] for key in data}
    t = _data[keys.split(', ')[0]]
    key = (rows / height, cols / width, str(t.device))
    scale = _SCALES.get(key)        # (a per-call host -> device copy of two floats from pageable memory is a synchronous copy)
    if scale is None:
        if len(_SCALES) > 256:
            _SCALES.clear()
        scale = _SCALES[key] = torch.tensor(key[:2], dtype=torch.float32, device=t.device).view(1, 1, 2)
    for key in keys.split(', '):
        _data[key] = _data[key] * scale
    return _data


class _HostStagedWork(object):
    """all-reduce of a GPU tensor under a backend that only moves host memory (the gloo backend of the CPU / single-GPU tests):
    copy to the host, reduce there, copy back on wait().  RCCL (backend "nccl") never takes this path."""

    def __init__(self, t, group):
        self.t, self.h = t, t.cpu()
        self.work = dist.all_reduce(self.h, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def wait(self):
        self.work.wait()
        self.t.copy_(self.h)


class DataParallelRCCL(nn.Module):
    """Gradient-averaging data parallelism over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on CPU)."""

    def __init__(self, module, process_group=None, bucket_bytes=BUCKET_BYTES):
        nn.Module.__init__(self)
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg)
        self.bucket_bytes = bucket_bytes
        self._staged = dist.get_backend(self.pg) == 'gloo'
        # identical replicas: rank 0's parameters and buffers win (the reference broadcasts GPU0's every step)
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                if t.is_cuda and self._staged:
                    h = t.detach().cpu()
                    dist.broadcast(h, 0, group=self.pg)
                    t.copy_(h)
                else:
                    dist.broadcast(t, 0, group=self.pg)
        self._build_buckets()
        self._pending = False
        for p in self._params:
            p.register_post_accumulate_grad_hook(self._late_hook)
        # modules whose backward produces gradients layer by layer announce them early (overlap with compute) and may write them
        # straight into the flat buckets (grad_buffer_hook: no copy into the bucket)
        for m in module.modules():
            if hasattr(m, 'grad_ready_hook'):
                m.grad_ready_hook = self._early_hook
                m.grad_buffer_hook = self._grad_buffer

    tune_synced = None
    # training steps (forward calls in train mode with gradients enabled, or replayed plans; counted per wrapper, identical on every rank) at which the ranks adopt rank 0's measured algorithm choices: after 3 whole
    # steps, then at a thinning schedule that picks up the shapes a multi-scale run visits later, then every 4096 calls.  Rank 0's entries
    # win, entries only this rank has (a shape rank 0 has not met) stay; plans are rebuilt only where something changed.
    SYNC_TUNE_CALLS = (4, 16, 64, 256, 1024, 4096)
    _calls = 0

    def _sync_tune(self):
        """Each rank times its kernels itself during the first steps at a new input shape; near-ties resolve differently from rank
        to rank, and the step time of the job is the slowest rank's.  One broadcast of rank 0's table makes the plans identical.
        Every rank reaches the collective (forward() triggers it on its own call count); whatever can fail locally happens outside
        it and degrades to "keep my own choices"."""
        import _hip
        dev = next((p.device for p in self._params if p.is_cuda), None) or next((p.device for p in self._params), None)
        first = dist.get_rank(self.pg) == 0
        src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0      # (None = the default group: rank 0 is rank 0)
        table = None
        if first:
            try:
                table = _hip.export_tune()
            except Exception as e:
                logging.warning('autotune choices not exported: %s' % e)
        box = [table]
        if self._staged or dev is None or dev.type != 'cuda':
            dist.broadcast_object_list(box, src=src, group=self.pg)
        else:
            dist.broadcast_object_list(box, src=src, group=self.pg, device=dev)
        if box[0] is None or dev is None:
            return
        if not first:
            try:
                _hip.import_tune(box[0], dev, merge=True)
            except Exception as e:      # never fatal: this rank keeps its own choices
                logging.warning('autotune choices not adopted: %s' % e)
                return
        self.tune_synced = len(box[0])

    def _all_reduce(self, t):
        if t.is_cuda and self._staged:
            return _HostStagedWork(t, self.pg)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _sum_small(self, t):
        if self.world <= 1:
            return False
        self._all_reduce(t).wait()
        return True

    # ---- bucketing: reverse registration order ~ the order in which backward produces gradients.  Every flat bucket ends
    # with one flag per parameter ("this rank produced a gradient for it"): after the sum a parameter nobody touched keeps
    # grad None on every rank, and a parameter only SOME ranks touched gets the averaged gradient on all of them.
    def _build_buckets(self):
        self._params = [p for p in self.module.parameters() if p.requires_grad]
        self._where = {}
        self._buckets = []
        cur, size = [], 0
        # Weights first, in reverse registration order ~ the order in which backward produces them; everything one-dimensional (BatchNorm
        # affine parameters, biases: 0.1 % of the bytes) goes into ONE last bucket.  The Darknet backward hands the affine gradients of
        # ALL layers out together after its last layer (one conversion launch, model.train_graph._darknet_bwd): mixed into the weight
        # buckets they held every bucket back until the end of backward, and no all-reduce overlapped anything.
        small = [p for p in self._params if p.dim() <= 1]
        weights = list(reversed([p for p in self._params if p.dim() > 1]))
        # a plugin that knows the order in which its backward finishes its weight gradients says so (the passthrough branch of Darknet is
        # registered before layers3 but finished after layers2: in registration order its bucket would hold six others back); the order
        # is a property of the module's code, identical on every rank
        told = []
        for m in self.module.modules():
            f = getattr(m, 'backward_param_order', None)
            if callable(f):
                told += [p for p in f() if p.requires_grad]
        if told and len({id(p) for p in told}) == len(told) and {id(p) for p in told} <= {id(p) for p in weights}:
            rest = {id(p) for p in told}
            weights = told + [p for p in weights if id(p) not in rest]
        for p in weights:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                self._buckets.append(cur)
                cur, size = [], 0
        if cur:
            self._buckets.append(cur)
        if small:
            self._buckets.append(list(reversed(small)))
        self._flat = []
        for bi, bucket in enumerate(self._buckets):
            off = 0
            for p in bucket:
                self._where[id(p)] = (bi, off)
                off += p.numel()
            self._flat.append(torch.zeros(off + len(bucket), dtype=bucket[0].dtype, device=bucket[0].device))
        self._reset()

    _views = None
    _had = frozenset()

    def _reset(self):
        if self._views is None:
            self._views = {}
        self._ready = [0] * len(self._buckets)
        self._done = set()
        self._works = [None] * len(self._buckets)
        self._next = 0          # buckets are launched strictly in index order: every rank issues the SAME collective sequence

    def _start(self):
        """First gradient event of a backward pass.  Who accumulates is decided HERE, at backward time (Train.iterate runs the forward
        before optimizer.zero_grad(), train.py:344-351: a snapshot taken in forward() would see last step's gradients on every parameter
        and switch the in-place bucket path off for good): a parameter whose .grad is set now keeps it, and a kept gradient that is a
        bucket slice of the previous pass (see _finalize; also a second backward without a forward in between: two outputs,
        retain_graph) becomes a private copy before this pass's _fill rewrites the bucket."""
        if not self._pending:
            self._pending = True
            self._count_step()          # (before this backward's first all-reduce: every rank's collective sequence is [tune broadcast,] bucket 0, bucket 1, ...)
            had = set()
            for q in self._params:
                if q.grad is not None:
                    if q.grad is self._views.get(id(q)):
                        q.grad = q.grad.clone()
                    had.add(id(q))
            self._had = had
            self._views = {}
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _grad_buffer(self, p):
        """Where the averaged gradient of `p` will live: its slice of the flat bucket.  A backward that writes the gradient there
        (model.train_graph) saves the copy into the bucket; None = not this time (an accumulating caller, a moved module, a
        parameter this wrapper does not reduce)."""
        w = self._where.get(id(p))
        if w is None or id(p) in self._done:
            return None
        self._start()
        if id(p) in self._had or p.grad is not None:
            return None
        bi, off = w
        flat = self._flat[bi]
        if flat.device != p.device or flat.dtype != p.dtype:
            return None
        return flat[off:off + p.numel()].view_as(p)

    def _launch_bucket(self, bi):
        """Zero the slots (and flags) of parameters without a gradient this step, then start the bucket's all-reduce.  EVERY
        rank launches EVERY bucket exactly once per backward, so the collective sequences of the ranks always match."""
        flat, bucket = self._flat[bi], self._buckets[bi]
        base = flat.numel() - len(bucket)
        if self._ready[bi] == len(bucket):           # the common case: one fill kernel for all flags
            flat[base:].fill_(1.0)
        else:
            for k, p in enumerate(bucket):
                if id(p) in self._done:
                    flat[base + k] = 1.0
                else:
                    _, off = self._where[id(p)]
                    flat[off:off + p.numel()].zero_()
                    flat[base + k] = 0.0
        self._works[bi] = self._all_reduce(flat)

    def _fill(self, p, g):
        bi, off = self._where[id(p)]
        if self._flat[bi].device != g.device:       # the module moved (train.py:423-432: .cpu() for evaluation, .cuda() to resume)
            self._flat[bi] = self._flat[bi].to(g.device)
        slot = self._flat[bi][off:off + p.numel()]
        if g.data_ptr() != slot.data_ptr() or g.numel() != slot.numel() or not g.is_contiguous():      # (else: written in place through _grad_buffer)
            slot.copy_(g.reshape(-1))
        self._done.add(id(p))
        self._ready[bi] += 1
        while self._next < len(self._buckets) and self._ready[self._next] == len(self._buckets[self._next]):
            self._launch_bucket(self._next)        # a complete bucket waits for its incomplete predecessors (launched by _finalize)
            self._next += 1

    def _early_hook(self, p, g):
        """Called from inside a module's backward with the finished gradient of parameter p."""
        if id(p) not in self._where or id(p) in self._done:
            return
        self._start()
        if id(p) in self._had and p.grad is not None:
            g = p.grad + g          # the caller accumulates: autograd will add g to p.grad AFTER this backward function returns; reduce kept + g
        self._fill(p, g)

    def _late_hook(self, p):
        if id(p) in self._done:
            return
        self._start()
        self._fill(p, p.grad)

    def _finalize(self):
        inv = 1.0 / self.world
        try:
            for bi in range(self._next, len(self._buckets)):
                # some parameters of this bucket (or of an earlier one) got no gradient on this rank: reduce it anyway, in order
                bucket = self._buckets[bi]
                dev = next((p.grad.device for p in bucket if p.grad is not None), self._flat[bi].device)
                if self._flat[bi].device != dev:
                    self._flat[bi] = self._flat[bi].to(dev)
                self._launch_bucket(bi)
            self._next = len(self._buckets)
            self._wait_all()
            for bi, bucket in enumerate(self._buckets):
                flat = self._flat[bi]
                base = flat.numel() - len(bucket)
                flags = flat[base:].tolist() if any(id(p) not in self._done for p in bucket) else None
                flat[:base].mul_(inv)
                for k, p in enumerate(bucket):
                    _, off = self._where[id(p)]
                    avg = flat[off:off + p.numel()].view_as(p)
                    # the averaged gradient IS the bucket slice (no copy back: ~70 launches and 2 x 200 MB per step).  The slice is
                    # rewritten by the next backward's _fill; optimizer.zero_grad() has dropped this reference by then, and a caller
                    # that keeps it (gradient accumulation) gets a private copy at the next forward()
                    if p.grad is not None and id(p) in self._had:
                        p.grad.copy_(avg)                 # accumulating caller: the bucket held kept + this step's gradient (see _early_hook)
                    elif p.grad is not None or (flags is not None and flags[k] > 0):      # (second case: only another rank produced a gradient for it)
                        p.grad = avg
                        self._views[id(p)] = avg
        finally:
            self._pending = False
            self._reset()

    def _tick(self, training=False):
        """Start of a step (forward() or a StepPlan's graph_begin): clean bookkeeping and - for TRAINING steps only - the step count and the
        tune exchange it triggers."""
        # a backward that raised leaves the bookkeeping half-filled: start every step from a clean slate
        self._pending = False
        self._reset()
        if training:
            self._count_step()

    def _count_step(self):
        """One training step begins its collective sequence here (the first gradient event of a hook-path backward, or a StepPlan's graph_begin).  A forward
        call is NOT a step: an evaluation / no_grad call through the wrapper (rank 0 alone summarising, train.py:209; an eval pass between epochs) is
        rank-local and must not advance the schedule below, or that rank would enter the broadcast one step before the others do while they issue a gradient
        all-reduce on the same group; an eval()-mode step that DOES back-propagate must advance it on every rank."""
        self._calls += 1
        if self.world > 1 and (self._calls in self.SYNC_TUNE_CALLS or self._calls % self.SYNC_TUNE_CALLS[-1] == 0):
            # the trigger is the wrapper's count of TRAINING steps and nothing else: every rank starts one training step per step, so every
            # rank reaches the broadcast at the same point of its collective sequence - whatever input shapes the ranks' own loaders drew
            # (with one process per GPU each rank's collate picks its multi-scale size itself, utils/data.py:135-141; a per-shape trigger
            # would put one rank into the broadcast while the others start a gradient all-reduce)
            self._sync_tune()

    # ---- the same protocol for a step that runs as captured hipGraph segments (model.train_graph.StepPlan): no autograd hooks fire, the
    # plan writes every gradient into its bucket slice and tells the wrapper where the segments end.  Collectives, their order and their
    # payloads are those of the hook path - a rank replaying a graph and a rank still warming up on another input size match.
    def graph_begin(self):
        self._tick(training=False)          # (the step is counted where the hook path counts it: in front of the first bucket's all-reduce, graph_launch)

    def graph_slot(self, p):
        w = self._where.get(id(p))
        if w is None:
            return None
        bi, off = w
        flat = self._flat[bi]
        if flat.device != p.device:
            flat = self._flat[bi] = flat.to(p.device)
        return flat[off:off + p.numel()].view_as(p)

    def graph_launch(self, lo, hi):
        """All-reduce buckets lo .. hi-1 (complete on this rank), in index order."""
        for bi in range(lo, hi):
            assert bi == self._next, (bi, self._next)
            if bi == 0:
                # the same point of the collective sequence as _start() on the hook path - behind the loss's positive-count all-reduce, in front of bucket 0 - so
                # a rank replaying graph segments and a rank on the hook path meet the tune broadcast at the same place
                self._count_step()
            flat = self._flat[bi]
            flat[flat.numel() - len(self._buckets[bi]):].fill_(1.0)
            self._works[bi] = self._all_reduce(flat)
            self._next = bi + 1

    overlap_stats = None      # set to a list: (event before, event after) the waits for the outstanding all-reduces, one pair per step

    def _wait_all(self):
        st = self.overlap_stats
        timed = st is not None and self._flat[0].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if timed:
            e1.record()
            st.append((e0, e1))
            del st[:-64]

    def graph_finish(self):
        """Wait for the buckets, average.  (The plan assigns the slices as the parameters' .grad.)"""
        assert self._next == len(self._buckets)
        self._wait_all()
        torch._foreach_mul_([flat[:flat.numel() - len(b)] for flat, b in zip(self._flat, self._buckets)], 1.0 / self.world)
        self._reset()

    def graph_views(self, params, grads):
        """The plan made these bucket slices the parameters' .grad: a later hook-path backward must treat them as such (see _start)."""
        for p in params:
            g = grads.get(id(p))
            if g is not None:
                self._views[id(p)] = g

    def forward(self, *args, **kwargs):
        # forward() only cleans the bookkeeping.  What counts as a TRAINING step (and drives the tune exchange) is a BACKWARD that all-reduces - counted in
        # _start(): an evaluation / no_grad / summary forward of one rank alone never reaches it, a differentiable eval()-mode step (frozen-BatchNorm
        # fine-tuning, model.train_graph.darknet_forward_eval_grad) does, and every rank runs one such backward per step
        self._tick(training=False)
        out = self.module(*args, **kwargs)
        # the region loss sums its positive count over THIS wrapper's group (model/__init__.py:162: mean over the positives of the
        # global batch): the reducer travels with the predictions (model.train_graph.DP_TAG)
        from model import train_graph
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            if isinstance(t, torch.Tensor):
                setattr(t, train_graph.DP_TAG, self._sum_small)
        return out


def sync_tune(dev=None, group=None):
    """Every rank adopts rank 0's measured algorithm table (replicated inference, or any model without the wrapper): the same collective
    DataParallelRCCL runs per new input shape.  Returns the number of entries adopted (None without torch.distributed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return None
    import _hip
    dev = torch.device('cuda', torch.cuda.current_device()) if (dev is None and torch.cuda.is_available()) else dev
    first = dist.get_rank(group) == 0
    src = dist.get_global_rank(group, 0) if group is not None else 0
    table = None
    if first:
        try:
            table = _hip.export_tune()
        except Exception as e:
            logging.warning('autotune choices not exported: %s' % e)
    box = [table]
    if dev is None or dev.type != 'cuda' or dist.get_backend(group) == 'gloo':
        dist.broadcast_object_list(box, src=src, group=group)
    else:
        dist.broadcast_object_list(box, src=src, group=group, device=dev)
    if box[0] is None or dev is None:
        return None
    if not first:
        _hip.import_tune(box[0], dev)
    return len(box[0])


def init_distributed():
    """One process per GPU: reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('Y2_DIST_BACKEND')      # tests: "gloo" with GPU tensors staged through the host (two ranks on ONE GPU)
        if torch.cuda.is_available():
            local = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            if backend in (None, '', 'nccl'):
                dist.init_process_group('nccl', device_id=torch.device('cuda', local))       # "nccl" IS RCCL on ROCm
            else:
                dist.init_process_group(backend)
        else:
            dist.init_process_group(backend or 'gloo')
    return world


def ensure_model(model_):
    """train.py:65-71: move to the GPU and wrap for multi-GPU training (callable like the module, .parameters(), .train())."""
    if torch.cuda.is_available():
        model_.cuda()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        logging.info('%d GPUs are used' % dist.get_world_size())
        model_ = DataParallelRCCL(model_)
    return model_


GRAPH = os.environ.get('Y2_TRAIN_GRAPH', '1') != '0'       # 0: every training step through autograd, launch by launch (A/B runs, per-kernel event tables)


class StepRunner(object):
    """The StepPlans (model.train_graph) of one model: one per input shape - per-GPU batch, input size (multi-scale training cycles
    through ten, utils/data.py:135-141), label form; its label buffers hold the largest box count met so far - all allocating from one graph memory pool, least recently used
    dropped beyond `MAX`.  step(data) returns what iterate returns, or None when this step cannot run as a plan (the caller then takes
    the autograd path)."""
    MAX = int(os.environ.get('Y2_TRAIN_PLANS', '24'))

    def __init__(self, inference, dp, anchors, hparam, threshold):
        import collections
        self.inference, self.dp = inference, dp
        self.anchors, self.hparam, self.threshold = anchors, dict(hparam), float(threshold)
        self.plans = collections.OrderedDict()
        self.warm = {}
        self.used = {}              # input shape -> (block, operand form) pairs the last eager pass of that shape read (StepPlan.used_last)
        self.pool = None
        self.arena = None           # torch.cuda.MemPool: the one activation arena of all plans (Y2_TRAIN_ARENA=0: a graph pool per capture generation, as before)
        self.shared_scope = {}      # with an arena: kernel scratch / gradient staging / constants of the captured steps, one set for all plans (StepPlan.scope)
        self.shared_ops = {}        # prepared GEMM-operand buffers, one set for the plans of all input shapes (they replay serially; each rewrites what it reads)
        self.broken = None          # a capture failed for a reason other than memory: no more captures (steps keep running as eager plans)
        self.eager_only = set()     # shapes whose capture failed
        self.captures = 0

    def _new_plan(self):
        from model import train_graph
        if ARENA and self.arena is None:
            try:
                self.arena = torch.cuda.MemPool()       # lives as long as this runner: the plans of all input shapes (eager passes and captures alike) allocate from it
            except Exception as e:                      # (an allocator without pools: every plan falls back to the graph-pool handle below)
                logging.warning('training-step arena unavailable (%s: %s)' % (type(e).__name__, e))
                self.arena = False
        if not self.arena and (self.pool is None or not any(p.ops is not None for p in self.plans.values())):
            # (a bare pool handle lives as long as a graph captured into it: once the last one is gone it is dead - torch asserts on reuse)
            self.pool = torch.cuda.graph_pool_handle()
        return train_graph.StepPlan(self.inference, self.anchors, self.hparam, self.threshold, dp=self.dp, pool=self.pool, shared=self.shared_ops, arena=self.arena or None,
                                    scope=self.shared_scope if self.arena else None)

    def reserve(self, data, plans=10, margin=1.03):
        """Size the activation arena ONCE for the largest problem shape the job will meet (`[data] sizes` lists them up front, /root/reference/config.ini:39;
        utils/data.py:135-141 changes the size every `maintain` batches): `data` = a batch of that shape (only shapes matter: NOTHING EXECUTES - weights,
        BatchNorm statistics and step counters are untouched).  `plans`: how many input sizes will follow (each keeps its static inputs and results).
          1. the step is captured into a throw-away pool: what that pool ends up holding is the step's footprint P (intermediates at their high-water mark,
             kernel scratch, GEMM operands); capture and pool are dropped;
          2. the arena gets ONE allocation of margin * P + the plans' own tensors, freed at once: a single segment, so the plans of every size carve their
             intermediates from it and the pieces coalesce again when a capture ends (separately grown segments never merge: ten sizes met in ascending order
             held 224 GiB, and 71 GiB with per-tensor growth from a first capture, for a 23 GiB working set);
          3. the step is captured once more, into the arena and the plans' shared scope: scratch and operand buffers exist at their largest size from here on.
        Without a reservation the arena grows as larger sizes arrive.  Returns the arena's bytes (None: no arena / not eligible / a capture failed)."""
        import gc

        import model
        from model import train_graph
        if not self.eligible(data) or not GRAPH:
            return None
        self._new_plan()          # (creates the arena)
        if not self.arena:
            return None
        n = data['yx_min'].shape[1]
        npad = 16
        while npad < n:
            npad *= 2
        # the step's small host-to-device constants are made on first use and cached - a copy no capture may contain (the eager warm-up passes of an ordinary
        # plan make them; here nothing runs first): anchors, the loss weights as the chain orders them
        dev = data['tensor'].device
        model._device_anchors(self.anchors, dev)
        base = ['foreground', 'background', 'center', 'size']
        for keys in (base, base + ['cls']):           # (single-class heads have no cls term: both forms of the weight vector, a few bytes each)
            if all(k in self.hparam for k in keys):
                train_graph._hparam_tensor(tuple(float(self.hparam[k]) for k in keys) + (0.0,) * (5 - len(keys)), dev)

        def pool_bytes(pool):
            return sum(seg['total_size'] for seg in torch.cuda.memory_snapshot() if tuple(seg.get('segment_pool_id', ())) == tuple(pool.id))

        def capture(arena, scope, shared):
            plan = train_graph.StepPlan(self.inference, self.anchors, self.hparam, self.threshold, dp=self.dp, shared=shared, arena=arena, scope=scope)
            plan._alloc(data, npad)
            plan._load(data)
            plan._capture()          # (under the data-parallel wrapper too: a capture issues no collective, it only notes where they belong)
            torch.cuda.synchronize()
            own = sum(t.numel() * t.element_size() for t in plan.static.values() if isinstance(t, torch.Tensor))
            return own
        try:
            probe = torch.cuda.MemPool()
            own = capture(probe, {}, {})
            gc.collect()
            need = pool_bytes(probe)
            del probe
            gc.collect()
            torch.cuda.empty_cache()
            params = sum(p.numel() * p.element_size() for p in self.inference.parameters())
            # (a plan's static inputs and the operand buffers are ordinary allocations outside the arena; inside it a plan keeps its result views: ~0.1 GB)
            slab_bytes = int(margin * need) + max(0, plans - 1) * (128 << 20) + params
            with torch.cuda.stream(train_graph._capture_stream(dev)), torch.cuda.use_mem_pool(self.arena):
                slab = torch.empty(slab_bytes, dtype=torch.uint8, device=dev)          # ONE segment ...
                del slab                                                                # ... of free arena from here on
            capture(self.arena, self.shared_scope, self.shared_ops)
            gc.collect()
        except Exception as e:          # never fatal: the arena then grows as the sizes arrive
            logging.warning('training-step arena not reserved (%s: %s)' % (type(e).__name__, str(e)[:200]))
            torch.cuda.synchronize()
            return None
        self.reserved = dict(step_bytes=need, slab_bytes=slab_bytes)
        return pool_bytes(self.arena)

    def _param_ids(self, dnn):
        """ids of every parameter slot of the module tree as it was when the plans were made (a module ADDED later has no plan-side gradient either way:
        the periodic full walk below catches it)."""
        self._walks = getattr(self, '_walks', 0) + 1
        if self._walks % 64 == 0 or getattr(self, '_mods_seen', None) is None:
            self._mods_seen = [m for m in dnn.modules() if m._parameters]
        return [id(p) for m in self._mods_seen for p in m._parameters.values()]

    def same(self, anchors, hparam, threshold):
        return anchors is self.anchors and dict(hparam) == self.hparam and float(threshold) == self.threshold

    def eligible(self, data):
        import _hip
        from model import resnet as _resnet
        from model import train_graph
        from model import yolo2 as _yolo2
        inf, x = self.inference, data.get('tensor')
        dnn = getattr(inf, 'dnn', None)
        if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dim() != 4 or x.shape[-1] % 32 or x.shape[-2] % 32:
            return False
        if not isinstance(inf, model.Inference) or not isinstance(dnn, (_yolo2.Darknet, _resnet.ResNet)) or not (inf.training and dnn.training):
            return False
        if _hip.DETERMINISTIC or train_graph.DEBUG_TAP is not None or not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return False
        if any(k not in data for k in ('yx_min', 'yx_max', 'cls')) or data['yx_min'].dim() != 3 or data['cls'].dim() not in (2, 3):
            return False
        ok = getattr(self, '_params_ok', None)
        if ok is not None:
            # a plan snapshots the parameter list and writes EVERY gradient: a parameter frozen after the first step (requires_grad = False:
            # fine-tuning schedules) or replaced by another tensor must drop the plans, not keep receiving gradients.  requires_grad is one
            # attribute read per parameter per step; the identity of the list is re-derived every 32 steps (a module-tree walk)
            # attribute read per parameter per step; the identity of the list is re-derived every step from the modules' own _parameters dicts (no generator
            # chain through named_modules: ~25 us for Darknet-19) - a replaced nn.Parameter must not leave even one step without its gradient
            ps = self._params_seen
            if not all(p.requires_grad for p in ps) or self._param_ids(dnn) != self._params_ids_seen:
                self.plans.clear()
                ok = None
        if ok is None:
            ps = self._params_seen = list(dnn.parameters())
            self._mods_seen = [m for m in dnn.modules() if m._parameters]
            self._params_ids_seen = self._param_ids(dnn)
            ok = all(p.requires_grad and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps)
            if ok and isinstance(dnn, _yolo2.Darknet) and not isinstance(dnn, _yolo2.Tiny):
                ok = train_graph._pad_layout(dnn) is None           # pruned widths run zero-padded through host-side glue: autograd path
            if ok and self.dp is not None:
                ok = all(id(p) in self.dp._where for p in ps) and len(ps) == len(self.dp._params)
            self._params_ok = ok
        return ok

    def step(self, data):
        import _hip
        from model import train_graph
        if not self.eligible(data):
            return None
        x, cls = data['tensor'], data['cls']
        n = data['yx_min'].shape[1]
        shape = (tuple(x.shape), tuple(cls.shape[2:]), cls.dim())
        tail = (_hip.tune_epoch(), _hip.WINOGRAD, _hip.split_mode(), _hip.FORCE_ALGO, train_graph.GRAD_F43, train_graph.FUSE_CONV0, str(x.device))
        # ONE plan per shape: a batch with fewer boxes runs in the plan captured for more (zero rows are the collate function's own padding); a batch
        # with more boxes than any plan of its shape holds supersedes them (label rows: powers of two from 16)
        mine = [k for k in self.plans if k[:3] == shape]
        for k in [k for k in mine if k[4:] != tail or not self.plans[k].valid()]:
            del self.plans[k]                        # measured on another algorithm table, or the model's memory moved (.cpu() / .cuda() around an evaluation)
            self._params_ok = None
        mine = [k for k in self.plans if k[:3] == shape]
        fit = sorted((k for k in mine if k[3] >= n), key=lambda k: k[3])
        if fit:
            key = fit[0]
            plan = self.plans[key]
            self.plans.move_to_end(key)
        else:
            npad = 16
            while npad < n:
                npad *= 2
            for k in mine:
                del self.plans[k]
            key = shape + (npad,) + tail
            plan = self._new_plan()
            plan._alloc(data, npad)
            plan.calls = self.warm.get(shape, 0)      # the per-layer measurements depend on the shape, not on the box count
            plan.used_last = self.used.get(shape)     # ... and so do the operand forms the chosen algorithms read: a successor plan (more boxes, a new tune epoch) captures the pruned step too
            self.plans[key] = plan
            while len(self.plans) > self.MAX:
                self.plans.popitem(last=False)
        npad = key[3]
        self.warm[shape] = self.warm.get(shape, 0) + 1
        had_graph = plan.ops is not None
        out = plan.run(data, capture=GRAPH and not self.broken and shape not in self.eager_only)
        if plan.used_last:
            self.used[shape] = plan.used_last
        self.last = (npad, 'replay' if had_graph else ('capture' if plan.ops is not None else 'eager'))      # (tools/soak_multiscale.py reads it)
        if plan.capture_error is not None and shape not in self.eager_only:
            # the capture failed; the step itself ran (eagerly).  Out of memory: every captured step is dropped (their shared pool goes back to
            # the allocator) and this shape stays on eager launches; anything else: no more captures for this model.
            e = plan.capture_error
            logging.warning('training-step capture failed (%s: %s); %s' % (type(e).__name__, str(e)[:200], 'eager launches for this input shape' if isinstance(e, torch.cuda.OutOfMemoryError) else 'eager launches from here on'))
            self.eager_only.add(shape)
            if isinstance(e, torch.cuda.OutOfMemoryError):
                for k in [k for k in self.plans if self.plans[k] is not plan]:
                    del self.plans[k]
                torch.cuda.empty_cache()
            else:
                self.broken = '%s: %s' % (type(e).__name__, e)
        if not had_graph and plan.ops is not None:
            self.captures += 1
        if self.dp is not None:
            self.dp.graph_views(plan.params, plan.last_grads)
        return out


ARENA = os.environ.get('Y2_TRAIN_ARENA', '1') != '0'      # 0: no shared activation arena (A/B)
PLAN = os.environ.get('Y2_TRAIN_PLAN', '1') != '0'      # 0: iterate never uses StepPlans (not even eagerly): the reference's three autograd calls


def _runner(inference, anchors, hparam, threshold):
    dp = inference if isinstance(inference, DataParallelRCCL) else None
    inner = inference.module if dp is not None else inference
    r = inner.__dict__.get('_y2_step_runner')
    if r is None or r.dp is not dp or not r.same(anchors, hparam, threshold):
        r = StepRunner(inner, dp, anchors, hparam, threshold)
        inner.__dict__['_y2_step_runner'] = r
    return r


def reserve(inference, data, loss_hparam, threshold, anchors):
    """Before the first step of a multi-scale job: size the training steps' activation arena for the LARGEST input size (`data`: a batch of that shape, e.g. the
    first batch resized like the collate function does - nothing executes, StepRunner.reserve).  Optional: without it the arena grows as larger sizes arrive."""
    if not (PLAN and isinstance(data.get('tensor'), torch.Tensor) and data['tensor'].is_cuda):
        return None
    return _runner(inference, anchors, loss_hparam, threshold).reserve(data)


def iterate(inference, optimizer, data, loss_hparam, threshold, anchors, clip=None):
    """Body of Train.iterate (train.py:338-362): forward, region loss, weighted sum, backward, optional clip, step.
    When the step has a fixed problem shape on the GPU it runs as a StepPlan: the same kernels in the same order, issued from a captured
    hipGraph instead of ~270 Python -> ctypes launches under autograd (Y2_TRAIN_GRAPH=0 / Y2_TRAIN_PLAN=0 switch that off)."""
    out = None
    if PLAN and isinstance(data.get('tensor'), torch.Tensor) and data['tensor'].is_cuda:
        out = _runner(inference, anchors, loss_hparam, threshold).step(data)
    if out is None:
        tensor = data['tensor']
        pred = model._inference(inference, tensor)
        height, width = tensor.size()[-2:]
        rows, cols = pred['feature'].size()[-2:]
        loss, debug = model.loss(anchors, norm_data(data, height, width, rows, cols), pred, threshold)
        loss_total = model.weighted_total(loss, loss_hparam)      # = sum(loss[key] * loss_hparam[key] for key in loss), one launch
        optimizer.zero_grad()
        loss_total.backward()
        out = dict(pred=pred, loss=loss, loss_total=loss_total, debug=debug)
    # (a StepPlan leaves this step's gradients in .grad - written over, never added to, what was there: zero_grad + backward in one)
    if clip is not None:
        utils.optim.clip_grad_norm_(inference.parameters(), clip)     # fused form of nn.utils.clip_grad_norm (train.py:352-354)
    optimizer.step()
    return out
