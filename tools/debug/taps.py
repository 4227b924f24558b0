"""Per-block data gradients of the Darknet training backward against the fp64 oracle's (round-3 debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import model, _hip
from model import train_graph
from oracle import darknet as odark, loss as oloss, synth, head as ohead
from oracle.make_golden import NARROW
import test_gpu_round3 as T

EARLY = {'layers1.0': 6, 'layers1.2': 10, 'layers1.5': 6, 'layers1.16': 30}
ours = {}
def tap(name, a, b, c, d):
    if name.endswith(':in'):
        ours[name] = (a.clone(), None if b is None else b.clone(), None if c is None else (c[0].clone(),) + tuple(c[1:]), None if d is None else d.clone())
    else:
        ours[name] = (a.clone(), b.clone())


train_graph.DEBUG_TAP = tap
w = dict(NARROW); w['layers1.5'] = 8; w.update(EARLY)
sd = odark.init_state_dict(5, 20, seed=0, channels=w, head_scale=1 / 8.0, bn=False)
inf, anchors = T.build(sd, bn=False)
inf.train()
x = synth.images(3, 96, seed=1)
data = synth.norm_data(synth.labels(3, 96, 20, seed=2), 96, 96, 3, 3)
pred = model._inference(inf, x.to('cuda:0'))
loss, _ = model.loss(anchors, data, pred, 0.6)
model.weighted_total(loss, oloss.HPARAM).backward()
torch.cuda.synchronize()
# oracle with taps (post-activation outputs of every block) and their gradients
sd64 = {k: v.double().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
taps = {}
f = odark.forward(x.double(), sd64, training=True, taps=taps)
for t in taps.values():
    t.retain_grad()
lo, _ = oloss.loss(anchors.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}, ohead.decode(f, anchors.double()), 0.6)
oloss.total(lo).backward()
order = [i[0] for i in odark.LAYERS1 if i != 'M']
pooled_after = {odark.LAYERS1[i][0] for i in range(len(odark.LAYERS1) - 1) if odark.LAYERS1[i] != 'M' and odark.LAYERS1[i + 1] == 'M'}
for j in range(1, len(order)):
    prod, cons = order[j - 1], order[j]
    if prod in pooled_after or cons not in ours:
        continue
    dx = ours[cons][1].permute(0, 3, 1, 2).cpu()
    ref = taps[prod].grad
    c = ref.shape[1]
    print('d loss / d output of %-10s (= dx of %-10s): err %.1e, pad channels max %.1e, shapes %s vs %s' % (prod, cons, T.rel(dx[:, :c], ref), dx[:, c:].abs().max().item() if dx.shape[1] > c else 0.0, tuple(dx.shape), tuple(ref.shape)))

# ---- block layers1.9: inputs of its y2_bn_act_bwd and the result, against the oracle and against a CPU evaluation of the same formula
z, shift, sf, sp = ours['layers1.9:in']
dz = ours['layers1.9'][0]
y_ref = taps['layers1.9']                      # leaky(conv + bias)
u_ref = torch.where(y_ref > 0, y_ref, y_ref / 0.1)
z_ours = z.permute(0, 3, 1, 2).cpu().double()
b = sd64['layers1.9.conv.bias'].detach().view(1, -1, 1, 1)
print('z(1.9) + bias vs oracle pre-activation: err %.1e; bias tensor equal: %s; sf meta %s sp %s' % (T.rel(z_ours + b, u_ref.detach()), torch.equal(shift.cpu(), sd['layers1.9.conv.bias']), sf[1:], None if sp is None else tuple(sp.shape)))
g = sf[0].permute(0, 3, 1, 2).cpu().double()
print('g vs oracle dL/dy(1.9): err %.1e' % T.rel(g, taps['layers1.9'].grad))
dz_formula = g * torch.where(z_ours + b > 0, 1.0, 0.1)
print('dz(1.9) kernel vs formula on ITS inputs: err %.1e;  vs oracle: %.1e' % (T.rel(dz.permute(0, 3, 1, 2).cpu(), dz_formula), T.rel(dz.permute(0, 3, 1, 2).cpu(), taps['layers1.9'].grad * torch.where(y_ref > 0, 1.0, 0.1))))
# ---- are the consumers of dz(1.9) right on THEIR inputs?
W = sd['layers1.9.conv.weight'].double()[:, :, 0, 0]              # [co=8][ci=16]
dzc = dz.cpu().double()                                            # [B,12,12,8]
dx19 = ours['layers1.9'][1].cpu().double()                        # [B,12,12,16]
print('dx(1.9) kernel vs W^T dz on its inputs: err %.1e' % T.rel(dx19, torch.einsum('bhwo,oi->bhwi', dzc, W)))
y18 = taps['layers1.8'].detach().permute(0, 2, 3, 1)               # oracle activation (ours matches it in the forward)
gw = dict(inf.dnn.named_parameters())['layers1.9.conv.weight'].grad.cpu().double()[:, :, 0, 0]
print('dW(1.9) kernel vs dz^T y(1.8): err %.1e;  oracle dW vs the same: %.1e' % (T.rel(gw, torch.einsum('bhwo,bhwi->oi', dzc, y18)), T.rel(sd64['layers1.9.conv.weight'].grad[:, :, 0, 0], torch.einsum('bhwo,bhwi->oi', dzc, y18))))
print('oracle: dL/dy(1.9) * act-derivative -> W^T gives dL/dy(1.8)? err %.1e' % T.rel(torch.einsum('bohw,oi->bihw', taps['layers1.9'].grad * torch.where(taps['layers1.9'] > 0, 1.0, 0.1).double(), W), taps['layers1.8'].grad))
print('mask agreement: %d of %d elements differ; |u| at the differing ones: %s' % (((z_ours + b > 0) != (y_ref > 0)).sum().item(), y_ref.numel(), u_ref[(z_ours + b > 0) != (y_ref > 0)].abs().detach()[:5].tolist()))
Bo = taps['layers1.9'].grad * torch.where(y_ref > 0, 1.0, 0.1)
d = (dz_formula - Bo).abs()
i = d.argmax()
gr = taps['layers1.9'].grad
print('worst element %d of %s: |diff| %.3e; ours g %.6e mask %.1f (z+b %.6e); oracle grad %.6e mask %.1f (y %.6e); rms(grad) %.3e rms(Bo) %.3e max|g-grad| %.3e; #|diff|>1e-3 rms: %d' % (
    i.item(), tuple(d.shape), d.flatten()[i].item(), g.flatten()[i].item(), (1.0 if (z_ours + b).flatten()[i] > 0 else 0.1), (z_ours + b).flatten()[i].item(),
    gr.flatten()[i].item(), (1.0 if y_ref.flatten()[i] > 0 else 0.1), y_ref.flatten()[i].item(), gr.pow(2).mean().sqrt().item(), Bo.pow(2).mean().sqrt().item(), (g - gr).abs().max().item(),
    (d > 1e-3 * Bo.pow(2).mean().sqrt()).sum().item()))
