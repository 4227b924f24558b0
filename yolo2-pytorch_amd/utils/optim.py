"""`utils.optim` — fused multi-tensor optimizers for the ini lambda (config.ini:72, evaluated at train.py:270).

    optimizer = lambda params, lr: utils.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-8)
    optimizer = lambda params, lr: utils.optim.SGD(params, lr, momentum=0.9)

Drop-in subclasses of torch.optim.Optimizer (same constructor arguments, `param_groups`, `state_dict()` layout with
`momentum_buffer` / `exp_avg`, `exp_avg_sq`, `step`, so checkpoints and lr schedulers — config.ini:75 — interchange with
torch.optim.SGD / Adam).  `step()` is one y2_opt_sgd / y2_opt_adam launch per 48 parameter tensors instead of one or more
kernels per tensor; `clip_grad_norm_` is the fused form of `nn.utils.clip_grad_norm` (train.py:352-354) without a host
synchronisation.  GPU fp32 parameters only — anything else raises (no CPU fallback).
"""
import ctypes

import torch

import _hip


def _check_tensor(t, what):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError('utils.optim: %s must be a contiguous fp32 GPU tensor (got %s, %s)' % (what, t.device, t.dtype))


def _launch(fn, rows, *args):
    """rows: list of (param, grad, state1, state2) tensors (None allowed for states) -> calls fn(table, count, *args, stream) in
    chunks of Y2_OPT_MAX_TENSORS."""
    st = _hip.stream()
    for lo in range(0, len(rows), _hip.OPT_MAX_TENSORS):
        chunk = rows[lo:lo + _hip.OPT_MAX_TENSORS]
        table = (_hip.OptTensor * len(chunk))()
        for e, (p, g, s1, s2) in zip(table, chunk):
            e.param = p.data_ptr() if p is not None else None
            e.grad = g.data_ptr()
            e.state1 = s1.data_ptr() if s1 is not None else None
            e.state2 = s2.data_ptr() if s2 is not None else None
            e.numel = g.numel()
        _hip.check(fn(table, len(chunk), *args, st), fn.__name__)


def _grad(p):
    g = p.grad
    if g.is_sparse:
        raise RuntimeError('utils.optim does not support sparse gradients')
    _check_tensor(p.data, 'a parameter')
    if g.dtype != torch.float32 or not g.is_contiguous():
        g = g.float().contiguous()
        p.grad = g
    _check_tensor(g, 'a gradient')
    return g


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD semantics (momentum, dampening, weight_decay, nesterov) in one fused launch per 48 tensors."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError('invalid hyper-parameter')
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _hip.lib()
        written = []
        for group in self.param_groups:
            fresh, seasoned = [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                g = _grad(p)
                written.append(p)
                buf = None
                first = False
                if group['momentum'] != 0:
                    state = self.state[p]
                    buf = state.get('momentum_buffer')
                    if buf is None:
                        buf = state['momentum_buffer'] = torch.empty_like(p.data, memory_format=torch.contiguous_format)
                        first = True
                (fresh if first else seasoned).append((p.data, g, buf, None))
            for rows, first in ((fresh, 1), (seasoned, 0)):
                if rows:
                    _launch(L.y2_opt_sgd, rows, group['lr'], group['momentum'], group['dampening'], group['weight_decay'],
                            int(group['nesterov']), first)
        if written:
            _hip.wrote(written)      # parameters were written through raw pointers: advance their version counters (packed-weight caches key on them)
        return loss


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas, eps, L2 weight_decay; amsgrad is not implemented) in one fused launch per 48 tensors."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError('utils.optim.Adam: amsgrad is not implemented')
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError('invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _hip.lib()
        written = []
        for group in self.param_groups:
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                g = _grad(p)
                written.append(p)
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = torch.tensor(0.0)          # host counter, torch.optim's layout
                    state['exp_avg'] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                state['step'] += 1
                by_step.setdefault(int(state['step']), []).append((p.data, g, state['exp_avg'], state['exp_avg_sq']))
            b1, b2 = group['betas']
            for step, rows in by_step.items():
                _launch(L.y2_opt_adam, rows, group['lr'], b1, b2, group['eps'], group['weight_decay'], step)
        if written:
            _hip.wrote(written)
        return loss


def clip_grad_norm_(parameters, max_norm):
    """Fused L2 `nn.utils.clip_grad_norm` (train.py:352-354): two launches per 48 tensors, no host synchronisation.  Returns the
    total norm as a 0-d GPU tensor (fp64)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    rows = [(None, _grad(p), None, None) for p in parameters if p.grad is not None]
    if not rows:
        return torch.zeros((), dtype=torch.float64)
    L = _hip.lib()
    sumsq = torch.zeros((), dtype=torch.float64, device=rows[0][1].device)
    _launch(L.y2_opt_grad_sumsq, rows, ctypes.c_void_p(sumsq.data_ptr()))
    _launch(L.y2_opt_clip_grads, rows, ctypes.c_void_p(sumsq.data_ptr()), float(max_norm))
    return sumsq.sqrt()


clip_grad_norm = clip_grad_norm_
