"""Adam over flat parameter arenas - one fused launch per step (reads p,g,m,v / writes p,m,v:
7 x 108 MB, HBM-bound) instead of torch.optim.Adam's per-tensor op chain.

Same surface as the calls the reference makes (``torch.optim.Adam(params, lr, betas)`` at
base_adaptor.py:126; ``zero_grad(); loss.backward(); step()`` at dynaboa_benchmark.py:149-151) and
the same update rule (no weight decay / amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
from __future__ import annotations

import torch

from . import _lib
from ._abi import check
from .hmr import stream_of


def materialize_grad(p):
    """The full gradient of a parameter whose last second-order accumulation was deferred to Adam.step (p.grad - alpha * h)."""
    pend = getattr(p, "_so_pending", None)
    if pend is None:
        return p.grad
    h, alpha = pend[0], pend[1]
    return p.grad - alpha * h


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        for p in self.params:
            if p.dtype != torch.float32 or p.numel() % 4 or not p.is_contiguous():
                raise ValueError("dynaboa_amd.optim.Adam expects contiguous fp32 arenas with numel % 4 == 0")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps)]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if getattr(p, "_so_pending", None) is not None:
                p._so_pending = None
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        g0 = self.param_groups[0]
        lr, (b1, b2), eps = g0["lr"], g0["betas"], g0["eps"]
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
            st["step"] += 1
            t = st["step"]
            step_size = lr / (1.0 - b1 ** t)
            bc2_sqrt = (1.0 - b2 ** t) ** 0.5
            g = p.grad.contiguous()
            pend = getattr(p, "_so_pending", None)
            if pend is not None:
                # second order: the gradient is still g - alpha * h (MAML deferred its last accumulation, maml.py) - one launch
                h, alpha = pend[0], pend[1]
                # the pair belongs to the gradient autograd left in .grad at the end of that backward (maml.py notes the tensor and
                # its version in a post-accumulate hook - the parameter may legitimately receive several accumulations in one
                # backward).  An in-place edit since (gradient clipping, a second backward) or a replaced .grad would be combined
                # with a stale H v: say so loudly (use dynaboa_amd.optim.materialize_grad(p) before editing gradients, or
                # MAML(defer_accumulate=False))
                seen = getattr(p, "_so_grad_seen", None)
                # (no hook record - .grad assigned by hand, a torch without post-accumulate hooks: fall back to the version counter,
                # which is 0 for a gradient autograd has accumulated exactly once and nobody has touched since)
                stale = (seen != (id(p.grad), p.grad._version)) if seen is not None else (p.grad._version != 0)
                if stale:
                    import warnings
                    warnings.warn("second-order gradient: .grad was modified or replaced after backward() while its last accumulation "
                                  "(v - lr * H v) was still deferred to Adam.step", RuntimeWarning, stacklevel=2)
                p._so_pending = None
                check(lib.dyb_adam_step_accum(p.data_ptr(), g.data_ptr(), h.data_ptr(), float(alpha), st["exp_avg"].data_ptr(),
                                              st["exp_avg_sq"].data_ptr(), b1, b2, step_size, bc2_sqrt, eps, p.numel(), stream_of(p)),
                      "dyb_adam_step_accum")
                torch.autograd.graph.increment_version(p)
                continue
            check(lib.dyb_adam_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                    b1, b2, step_size, bc2_sqrt, eps, p.numel(), stream_of(p)), "dyb_adam_step")
            torch.autograd.graph.increment_version(p)      # (the library's kernels write p in place behind autograd's back)


def ema_update(teacher_params, model_params, alpha: float):
    """teacher = alpha * teacher + (1 - alpha) * model  (reference base_adaptor.py:193-201)."""
    lib = _lib.load()
    for pt, p in zip(teacher_params, model_params):
        check(lib.dyb_ema_update(pt.data_ptr(), p.data_ptr(), float(alpha), pt.numel(), stream_of(pt)), "dyb_ema_update")
        torch.autograd.graph.increment_version(pt)
