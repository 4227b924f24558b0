"""Find an out-of-bounds writer in the training graph: every tensor train_graph allocates gets sentinel-filled guard zones, and the
guards of all live tensors are verified after EVERY library call (the wrapper of _hip.check knows the call's name)."""
import os, sys, weakref
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import model, _hip
from model import train_graph
from oracle import darknet as odark, loss as oloss, synth
from oracle.make_golden import NARROW
import test_gpu_round3 as T

G = 256
SENT = 12345.678
live = []
counter = [0]


def guarded_new(dev, *shape, dtype=torch.float32):
    n = 1
    for s in shape:
        n *= s
    if dtype != torch.float32:
        return torch.empty(*shape, dtype=dtype, device=dev)
    buf = torch.full((n + 2 * G,), SENT, dtype=torch.float32, device=dev)
    view = buf[G:G + n].view(*shape)
    counter[0] += 1
    live.append((weakref.ref(view), buf, n, tuple(shape), counter[0]))
    return view


reported = set()


def verify(what):
    torch.cuda.synchronize()
    keep = []
    for ref, buf, n, shape, ident in live:
        if ref() is None:
            continue
        keep.append((ref, buf, n, shape, ident))
        head, tail = buf[:G], buf[G + n:]
        bad_h, bad_t = (head != SENT).nonzero(), (tail != SENT).nonzero()
        if (len(bad_h) or len(bad_t)) and ident not in reported:
            reported.add(ident)
            print('GUARD VIOLATION after %s: tensor #%d shape %s: %d head / %d tail guard words overwritten (first tail idx %s, head idx %s)' %
                  (what, ident, shape, len(bad_h), len(bad_t), bad_t[:3].flatten().tolist(), bad_h[-3:].flatten().tolist()))
    live[:] = keep


orig_check = _hip.check


def checked(rc, what):
    orig_check(rc, what)
    verify(what)


train_graph._new = guarded_new
_hip.check = checked
EARLY = {'layers1.0': 6, 'layers1.2': 10, 'layers1.5': 6, 'layers1.16': 30}
train_graph.BWD_STREAMS = 1
w = dict(NARROW); w['layers1.5'] = 8; w.update(EARLY)
sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0, bn=False)
inf, anchors = T.build(sd, bn=False)
inf.train()
x = synth.images(3, 96, seed=1)
data = synth.norm_data(synth.labels(3, 96, 20, seed=2), 96, 96, 3, 3)
pred = model._inference(inf, x.to('cuda:0'))
print('forward done, %d guarded tensors' % counter[0])
loss, _ = model.loss(anchors, data, pred, 0.6)
model.weighted_total(loss, oloss.HPARAM).backward()
verify('end')
sd64, lo, stats, f = T.oracle_step(sd, x, data, anchors, True)
ours = dict(inf.dnn.named_parameters())
print('with guards: weight-grad errors:', ' '.join('%s=%.0e' % (k.replace('.conv.weight', '').replace('layers', 'L'), T.rel(ours[k].grad, v.grad)) for k, v in sd64.items() if v.requires_grad and k.endswith('conv.weight')))
