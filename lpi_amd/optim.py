"""SGD with momentum and weight decay on the HIP path (lpi_sgd_step), plus the cosine schedule — replaces optim.SGD(momentum=0.9, lr, weight_decay)
and optim.lr_scheduler.CosineAnnealingLR(T_max=epochs) as the reference builds them (retrieval/methods/sprompt.py:253-255), same arithmetic:
d = grad + wd * p; buf = d on the first step, momentum * buf + d after; p -= lr * buf.

The trainable set of the path is tiny (five DecomposedPrompt factors, 5 284 floats: sprompt.py:230-237), so the whole optimiser step is ONE launch
when the parameters are views of one flat buffer (``flatten``) and their gradients views of another — which is also what the data-parallel SUM
all-reduce wants (dp.Exchange.allreduce_grads takes the flat gradient as it is: no cat, no copy back).
"""
from __future__ import annotations

import math

import torch

from . import _lib


def flatten(params):
    """Re-seat the given leaf tensors (dict name -> tensor, or list) as views of ONE flat f32 buffer and allocate a flat gradient buffer of the
    same layout.  Returns (flat_param, flat_grad, grad_views): grad_views[i] is the slice of flat_grad with params[i]'s shape — hand these to
    functional.DecomposedPromptFn (grad_out=...) so that autograd's .grad tensors ARE those slices."""
    plist = list(params.values()) if isinstance(params, dict) else list(params)
    n = sum(p.numel() for p in plist)
    dev = plist[0].device
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    grad = torch.zeros(n, dtype=torch.float32, device=dev)
    o, views = 0, []
    with torch.no_grad():
        for p in plist:
            k = p.numel()
            flat[o:o + k].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + k].view(p.shape)
            views.append(grad[o:o + k].view(p.shape))
            o += k
    return flat, grad, views


class FlatSGD:
    """torch.optim.SGD(params, lr, momentum, weight_decay) for parameters that ``flatten`` laid out: step() is one lpi_sgd_step launch over the
    flat buffers when every .grad is the matching slice of the flat gradient, one launch per parameter otherwise (gradients that came from
    elsewhere).  param_groups[0]['lr'] is read at every step, so torch's lr schedulers and ``cosine_lr`` both work."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, flat=None, flat_grad=None, grad_views=None):
        self.params = list(params.values()) if isinstance(params, dict) else list(params)
        if flat is None:
            flat, flat_grad, grad_views = flatten(self.params)
        self.flat, self.flat_grad, self.grad_views = flat, flat_grad, grad_views
        self.buf = torch.zeros_like(flat)
        self.param_groups = [{"lr": float(lr), "momentum": float(momentum), "weight_decay": float(weight_decay), "params": self.params,
                              "initial_lr": float(lr)}]
        self.steps = 0

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def _is_flat(self):
        for p, v in zip(self.params, self.grad_views):
            g = p.grad
            if g is None or g.data_ptr() != v.data_ptr() or g.shape != v.shape or not g.is_contiguous():
                return False
        return True

    def _check_seating(self):
        """flatten() re-seated every parameter's .data as a view of self.flat; step() updates self.flat.  Anything that re-seats p.data afterwards
        (module.to(other device / dtype), .half(), load_state_dict(assign=True), a second flatten over the same tensors) would leave the optimiser
        updating a buffer the model no longer reads — a silent no-op training.  Checked at every step (five pointer compares); raises instead."""
        o = 0
        for p in self.params:
            k = p.numel()
            if p.data_ptr() != self.flat.data_ptr() + 4 * o or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FlatSGD: a parameter is no longer a view of the optimiser's flat buffer (its .data was re-seated after the optimiser "
                                   "was built: .to() / .half() / load_state_dict(assign=True) / a second flatten); build the optimiser after moving the module")
            o += k

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        s = torch.cuda.current_stream().cuda_stream
        first = 1 if self.steps == 0 else 0
        self._check_seating()
        if self._is_flat():
            _lib.call("lpi_sgd_step", self.flat.numel(), self.flat, self.flat_grad, self.buf, g["lr"], g["momentum"], g["weight_decay"], first, s)
        else:
            o = 0
            for p in self.params:
                k = p.numel()
                if p.grad is not None:
                    gr = p.grad.detach().contiguous().float()
                    _lib.call("lpi_sgd_step", k, self.flat[o:o + k], gr, self.buf[o:o + k], g["lr"], g["momentum"], g["weight_decay"], first, s)
                o += k
        self.steps += 1


def cosine_lr(base_lr: float, epoch: int, t_max: int, eta_min: float = 0.0) -> float:
    """Closed form of CosineAnnealingLR(T_max) after `epoch` scheduler steps (sprompt.py:254)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


class CosineLR:
    """optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max) for FlatSGD (torch's schedulers insist on a torch Optimizer): step() once per
    epoch, as sprompt.py:324 does."""

    def __init__(self, optimizer, T_max: int, eta_min: float = 0.0):
        self.opt, self.t_max, self.eta_min, self.epoch = optimizer, int(T_max), float(eta_min), 0
        self.base = [g["lr"] for g in optimizer.param_groups]

    def step(self):
        self.epoch += 1
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = cosine_lr(b, self.epoch, self.t_max, self.eta_min)

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]
