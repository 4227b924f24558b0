"""`detect` — detection post-filter of YOLOv2, MI355X-native mirror of detect.py:43-80.

`filter_visible` / `postprocess` keep the reference's signatures (one image, config-driven thresholds);
`postprocess_batch` is the device-resident form of the same a20 -> a21 -> a22 chain for a whole
batch: decode+softmax+max (y2_decode), threshold compaction (y2_filter_visible), rank + greedy NMS (y2_nms),
with no host round trip between the stages.  The OpenCV capture / drawing CLI of the reference (detect.py:83-214)
is out of scope.
"""
import torch

import _hip
import model
import utils
import utils.postprocess


def get_logits(pred):
    """detect.py:43-48."""
    if 'logits' in pred:
        return pred['logits'].contiguous()
    else:
        return torch.ones(*pred['iou'].size(), 1, device=pred['iou'].device)


def filter_visible_batch(iou, prob_cls, fix, thr):
    """iou, prob_cls [B,n] -> (count int32 [B], index int32 [B,n]); detect.py:53-62 for every image at once."""
    _hip.require_gpu(iou, prob_cls)
    iou, prob_cls = _hip.f32c(iou), _hip.f32c(prob_cls)
    B, n = iou.shape
    count = torch.empty(B, dtype=torch.int32, device=iou.device)
    index = torch.empty(B, max(n, 1), dtype=torch.int32, device=iou.device)
    _hip.check(_hip.lib().y2_filter_visible(_hip.ptr(iou), None, B, n, 1, int(bool(fix)), float(thr), _hip.ptr(count), _hip.ptr(index),
                                            _hip.ptr(prob_cls), None, _hip.stream()), 'y2_filter_visible')
    return count, index


def filter_visible(config, iou, yx_min, yx_max, prob):
    """detect.py:51-63, one image: iou [n], yx_min/yx_max [n,2], prob [n,C]."""
    _hip.require_gpu(iou, prob)
    iou, prob = _hip.f32c(iou).view(1, -1), _hip.f32c(prob)
    n, C = iou.size(1), prob.size(-1)
    fix = config.getboolean('detect', 'fix')
    thr = config.getfloat('detect', 'threshold_cls') if fix else config.getfloat('detect', 'threshold')
    dev = iou.device
    count = torch.empty(1, dtype=torch.int32, device=dev)
    index = torch.empty(1, max(n, 1), dtype=torch.int32, device=dev)
    prob_cls = torch.empty(1, max(n, 1), dtype=torch.float32, device=dev)
    cls = torch.empty(1, max(n, 1), dtype=torch.int32, device=dev)
    _hip.check(_hip.lib().y2_filter_visible(_hip.ptr(iou), _hip.ptr(prob.view(-1, C)), 1, n, C, int(fix), thr, _hip.ptr(count), _hip.ptr(index),
                                            _hip.ptr(prob_cls), _hip.ptr(cls), _hip.stream()), 'y2_filter_visible')
    idx = index[0, :int(count.item())].long()
    return (iou[0][idx], yx_min.view(-1, 2)[idx], yx_max.view(-1, 2)[idx], prob.view(-1, C)[idx], prob_cls[0][idx], cls[0][idx].long())


def _expand(iou, prob, yx_min, yx_max, cand, keep, keep_count, fix, threshold_cls):
    """y2_expand_classes: survivors of every image gathered and (fix) expanded into (box, class) detections.  iou [B,n], prob [B,n,C],
    yx_min / yx_max [B,n,2], cand int32 [B,n] or None, keep int32 [B,limit], keep_count int32 [B]."""
    B, limit = keep.shape
    n, C = iou.size(1), prob.size(-1)
    dev = iou.device
    new = lambda *s, **kw: torch.empty(*s, device=dev, **kw)
    k_iou, k_min, k_max = new(B, limit), new(B, limit, 2), new(B, limit, 2)
    if fix:
        e_min, e_max, e_score = new(B, limit * C, 2), new(B, limit * C, 2), new(B, limit * C)
        e_cls, e_count = new(B, limit * C, dtype=torch.int64), new(B, dtype=torch.int32)
    else:
        e_min = e_max = e_score = e_cls = e_count = None
    _hip.check(_hip.lib().y2_expand_classes(_hip.ptr(iou), _hip.ptr(prob), _hip.ptr(yx_min), _hip.ptr(yx_max), _hip.ptr(cand), _hip.ptr(keep), _hip.ptr(keep_count),
                                            B, n, C, limit, float(threshold_cls), _hip.ptr(k_iou), _hip.ptr(k_min), _hip.ptr(k_max),
                                            _hip.ptr(e_min), _hip.ptr(e_max), _hip.ptr(e_score), _hip.ptr(e_cls), _hip.ptr(e_count), _hip.stream()), 'y2_expand_classes')
    return k_iou, k_min, k_max, e_min, e_max, e_score, e_cls, e_count


def postprocess(config, iou, yx_min, yx_max, prob):
    """detect.py:66-80, one image: visibility filter -> NMS on objectness -> per-class expansion (`[detect] fix`) or arg-max class.
    Returns (iou, yx_min, yx_max, cls, score) or None when nothing survives."""
    iou, yx_min, yx_max, prob, prob_cls, cls = filter_visible(config, iou, yx_min, yx_max, prob)
    keep = utils.postprocess.nms(iou, yx_min, yx_max, config.getfloat('detect', 'overlap'))
    if not keep:
        return None
    fix = config.getboolean('detect', 'fix')
    k = len(keep)
    dev = iou.device
    keep_t = torch.tensor(keep, dtype=torch.int32, device=dev).view(1, k)
    count = torch.tensor([k], dtype=torch.int32, device=dev)
    k_iou, k_min, k_max, e_min, e_max, e_score, e_cls, e_count = _expand(_hip.f32c(iou).view(1, -1), _hip.f32c(prob).view(1, iou.numel(), -1), _hip.f32c(yx_min).view(1, -1, 2),
                                                                        _hip.f32c(yx_max).view(1, -1, 2), None, keep_t, count, fix, config.getfloat('detect', 'threshold_cls'))
    if fix:
        m = int(e_count.item())
        return k_iou[0], e_min[0, :m], e_max[0, :m], e_cls[0, :m], e_score[0, :m]
    return k_iou[0], k_min[0], k_max[0], cls[keep_t[0].long()], k_iou[0]


def detect_batch(feature_nhwc, anchors, fix=False, threshold=0.3, threshold_cls=0.005, overlap=0.45, limit=200):
    """Device-resident a8 -> a19 -> a20 -> a21: head image [B,rows,cols,A*(5+C)] -> dict of GPU tensors:
    decoded boxes/probabilities, per-image candidate list (count, index) and NMS survivors (keep = positions in the
    candidate list, keep_count).  No host synchronisation."""
    A = anchors.size(0)
    d = model.decode(feature_nhwc, anchors, A, want_prob=True)
    B = feature_nhwc.size(0)
    n = d['iou'].numel() // B
    iou = d['iou'].view(B, n)
    thr = threshold_cls if fix else threshold
    count, index = filter_visible_batch(iou, d['prob_cls'].view(B, n), fix, thr)
    keep, keep_count = utils.postprocess.nms_batch(iou, d['yx_min'].view(B, n, 2), d['yx_max'].view(B, n, 2), count, overlap, limit, cand=index)
    d.update(count=count, index=index, keep=keep, keep_count=keep_count)
    return d


def postprocess_batch(d, fix=False, threshold_cls=0.005, to_host=False):
    """a22 (detect.py:69-79) for every image of a detect_batch result: ONE launch (y2_expand_classes: gather of the survivors and, with
    `fix`, their expansion into (box, class) detections in row-major order) and one host synchronisation for the counts.
    Returns a list (per image) of None or (iou, yx_min, yx_max, cls, score) GPU tensors (views of the batch's result buffers).
    to_host: the tuples hold CPU tensors - the whole batch's result buffers cross PCIe in ONE copy per buffer (six copies, one synchronisation) and are sliced
    on the host, instead of five small copies and a synchronisation per image (what calling .cpu() on the GPU views costs: the consumer of the reference's
    postprocess works on the host, utils/postprocess.py:34-49 returns a Python list)."""
    keep = d['keep']
    B, limit = keep.shape
    n = d['iou'].numel() // B
    k_iou, k_min, k_max, e_min, e_max, e_score, e_cls, e_count = _expand(d['iou'].view(B, n), d['prob'].view(B, n, -1), d['yx_min'].view(B, n, 2), d['yx_max'].view(B, n, 2),
                                                                        d['index'], keep, d['keep_count'], fix, threshold_cls)
    if to_host and fix:
        bufs = [t.to('cpu', non_blocking=True) for t in (torch.stack([d['keep_count'], e_count]), k_iou, e_min, e_max, e_cls, e_score)]
        torch.cuda.current_stream().synchronize()
        counts, h_iou, h_min, h_max, h_cls, h_score = bufs[0].tolist(), bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]
        return [None if counts[0][b] == 0 else (h_iou[b, :counts[0][b]], h_min[b, :counts[1][b]], h_max[b, :counts[1][b]], h_cls[b, :counts[1][b]], h_score[b, :counts[1][b]]) for b in range(B)]
    counts = (torch.stack([d['keep_count'], e_count]) if fix else d['keep_count'].view(1, B)).tolist()      # the one host round trip
    out = []
    for b in range(B):
        k = counts[0][b]
        if k == 0:
            out.append(None)
        elif fix:
            m = counts[1][b]
            out.append((k_iou[b, :k], e_min[b, :m], e_max[b, :m], e_cls[b, :m], e_score[b, :m]))
        else:
            src = d['index'][b].long()[keep[b, :k].long()]
            out.append((k_iou[b, :k], k_min[b, :k], k_max[b, :k], d['cls'].view(B, n)[b][src].long(), k_iou[b, :k]))
    if to_host:
        out = [None if r is None else tuple(t.cpu() for t in r) for r in out]
    return out


class GraphedDetector(object):
    """hipGraph capture of the whole device-resident detect step (conv stack + decode + filter + NMS, ~30 launches) for a
    fixed input shape: one graph launch per batch instead of ~30 kernel launches and ~40 tensor allocations from Python.
    `run(x)` copies x into the static input buffer, replays the graph and returns the static result dict (overwritten
    by the next run).  `static_input=True` captures on `example` itself (no private copy): `run()` then re-reads that tensor.
    The graph bakes in the pointers of the packed / folded weights: after the parameters change (optimizer step, load_state_dict,
    a training forward) a replay would use stale weights, so `run` checks the plugin's cache key and refuses."""

    def __init__(self, dnn, anchors, example, fix=True, threshold=0.3, threshold_cls=0.005, overlap=0.45, limit=200, warmup=2, static_input=False, slot=0):
        """slot: detectors captured in different slots own different intermediate buffers (model.yolo2.Darknet.forward_nhwc) and may be
        replayed concurrently on different streams; detectors of one slot share them and must be replayed one after the other."""
        self.static_x = example if static_input else example.clone()
        self.dnn = dnn
        kw = dict(fix=fix, threshold=threshold, threshold_cls=threshold_cls, overlap=overlap, limit=limit)

        def step():
            with torch.no_grad():
                return detect_batch(dnn.forward_nhwc(self.static_x, slot) if slot else dnn.forward_nhwc(self.static_x), anchors, **kw)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):          # warm-up on a side stream: one-time attribute/symbol calls, plan + weight packing
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = step()
        # what the captured pointers were derived from: the tensors themselves (kept alive here) with their version counters - the check in
        # run() is one attribute read per tensor, no module-tree walk (it sits in the replay path)
        self._watched = [(t, t.data_ptr(), t._version) for t in list(dnn.parameters()) + list(dnn.buffers())]
        # the graph bakes in the addresses of the plan's intermediate buffers and scratch: hold the plan, so that an eviction from the
        # plugin's plan LRU (a 13th input shape) cannot hand that memory to somebody else while this graph can still be replayed
        plans = getattr(dnn, '_plans', None)
        self._plan = plans.latest() if plans is not None else None

    def run(self, x=None):
        for t, ptr, ver in self._watched:
            if t._version != ver or t.data_ptr() != ptr:
                raise RuntimeError('GraphedDetector: the parameters changed since capture (the graph holds the old packed weights); capture a new one')
        if x is not None and x.data_ptr() != self.static_x.data_ptr():
            self.static_x.copy_(x)
        self.graph.replay()
        return self.result
