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


def _expand_classes(iou, yx_min, yx_max, prob, threshold_cls):
    """The `fix` branch of detect.py:73-77: every (kept box, class) pair whose score = iou * prob exceeds the class threshold becomes
    a detection, in row-major (box, class) order.  Returns (yx_min, yx_max, cls, score) of the pairs."""
    score = iou.unsqueeze(-1) * prob
    box, cls = (score > threshold_cls).nonzero(as_tuple=True)
    return yx_min[box], yx_max[box], cls, score[box, cls]


def postprocess(config, iou, yx_min, yx_max, prob):
    """detect.py:66-80, one image: visibility filter -> NMS on objectness -> per-class expansion (`[detect] fix`) or arg-max class.
    Returns (iou, yx_min, yx_max, cls, score) or None when nothing survives."""
    iou, yx_min, yx_max, prob, prob_cls, cls = filter_visible(config, iou, yx_min, yx_max, prob)
    keep = utils.postprocess.nms(iou, yx_min, yx_max, config.getfloat('detect', 'overlap'))
    if not keep:
        return None
    keep = torch.tensor(keep, dtype=torch.long, device=iou.device)
    iou, yx_min, yx_max, prob, cls = iou[keep], yx_min[keep], yx_max[keep], prob[keep], cls[keep]
    if config.getboolean('detect', 'fix'):
        yx_min, yx_max, cls, score = _expand_classes(iou, yx_min, yx_max, prob, config.getfloat('detect', 'threshold_cls'))
        return iou, yx_min, yx_max, cls, score
    return iou, yx_min, yx_max, cls, iou


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


def postprocess_batch(d, fix=False, threshold_cls=0.005):
    """a22 (detect.py:69-79) for every image of a detect_batch result; one host sync.
    Returns a list (per image) of None or (iou, yx_min, yx_max, cls, score) GPU tensors."""
    B = d['keep'].size(0)
    n = d['iou'].numel() // B
    counts = d['keep_count'].tolist()
    out = []
    iou = d['iou'].view(B, n)
    mn, mx = d['yx_min'].view(B, n, 2), d['yx_max'].view(B, n, 2)
    prob = d['prob'].view(B, n, -1)
    for b in range(B):
        if counts[b] == 0:
            out.append(None)
            continue
        src = d['index'][b].long()[d['keep'][b, :counts[b]].long()]
        _iou, _mn, _mx, _prob = iou[b][src], mn[b][src], mx[b][src], prob[b][src]
        if fix:
            e_mn, e_mx, cls, score = _expand_classes(_iou, _mn, _mx, _prob, threshold_cls)
            out.append((_iou, e_mn, e_mx, cls, score))
        else:
            out.append((_iou, _mn, _mx, d['cls'].view(B, n)[b][src].long(), _iou))
    return out


class GraphedDetector(object):
    """hipGraph capture of the whole device-resident detect step (conv stack + decode + filter + NMS, ~30 launches) for a
    fixed input shape: one graph launch per batch instead of ~30 kernel launches and ~40 tensor allocations from Python.
    `run(x)` copies x into the static input buffer, replays the graph and returns the static result dict (overwritten
    by the next run).  `static_input=True` captures on `example` itself (no private copy): `run()` then re-reads that tensor.
    The graph bakes in the pointers of the packed / folded weights: after the parameters change (optimizer step, load_state_dict,
    a training forward) a replay would use stale weights, so `run` checks the plugin's cache key and refuses."""

    def __init__(self, dnn, anchors, example, fix=True, threshold=0.3, threshold_cls=0.005, overlap=0.45, limit=200, warmup=2, static_input=False):
        self.static_x = example if static_input else example.clone()
        self.dnn = dnn
        kw = dict(fix=fix, threshold=threshold, threshold_cls=threshold_cls, overlap=overlap, limit=limit)

        def step():
            with torch.no_grad():
                return detect_batch(dnn.forward_nhwc(self.static_x), anchors, **kw)
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
        self._versions = dnn._versions()

    def run(self, x=None):
        if self.dnn._versions() != self._versions:
            raise RuntimeError('GraphedDetector: the parameters changed since capture (the graph holds the old packed weights); capture a new one')
        if x is not None and x.data_ptr() != self.static_x.data_ptr():
            self.static_x.copy_(x)
        self.graph.replay()
        return self.result
