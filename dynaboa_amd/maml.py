"""MAML wrapper with the learn2learn surface the reference uses
(``l2l.algorithms.MAML(model, lr=fastlr, first_order=True)`` at reference base_adaptor.py:119;
``.clone()`` / ``.adapt(loss)`` at dynaboa_benchmark.py:136,140; semantics in SURVEY Appendix B).

Because the wrapped HMR keeps its 169 tensors in one arena, a learner is just "the module + one
fast-weight arena":  adapt() is one autograd.grad over the coarse engine node followed by ONE
fused ``p' = p - lr*g`` launch instead of 169 sub/mul pairs (18 % of the reference's CPU frame
time).

First order is what the reference runs.  Second order (``first_order=False``, learn2learn's
``create_graph=True`` through every inner step): the outer gradient of K inner steps is
v_k = (I - lr*H_k) v_{k+1}  with H_k the Hessian of the k-th lower-level loss at the k-th fast
weights, so ``adapt`` needs Hessian-vector products of the level's loss.  Two sources:
  * ``hvp_factory`` - exact, forward-over-reverse through the library's tangent passes
    (dynaboa_amd/hvp.py, csrc/hvp_engine.inc; the drivers' default ``--hvp exact``);
  * ``closure`` - the loss as a function of the learner; H v is then the central difference of two
    FIRST-order gradients,  H v ~ (g(theta + e v) - g(theta - e v)) / 2e,  e = fd_rel*|theta|/|v|
    (fd_rel = 1e-5, chosen by a sweep against the reference's second-order goldens - see
    MAML.fd_rel; ``--hvp fd``, and the fallback for levels the exact form does not cover).
``learner.adapt(loss, closure=..., hvp_factory=...)`` is the one extension over the learn2learn
signature; first-order calls stay ``adapt(loss)``."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._abi import check
from .hmr import stream_of


class _FastWeightStep(torch.autograd.Function):
    """out = p - lr * g.  First-order: the gradient flows to p unchanged and not into g."""

    @staticmethod
    def forward(ctx, p, g, lr):
        out = torch.empty_like(p)
        check(_lib.load().dyb_fastweight_update(p.data_ptr(), g.data_ptr(), out.data_ptr(), float(lr), p.numel(),
                                                stream_of(p)), "dyb_fastweight_update")
        return out

    @staticmethod
    def backward(ctx, d_out):
        return d_out, None, None


def _note_grad(p):
    """post-accumulate-grad hook of a parameter whose last second-order accumulation may be deferred: remember WHICH tensor
    autograd left in .grad and at which version, so that Adam.step can tell whether anybody edited or replaced it since."""
    g = p.grad
    p._so_grad_seen = None if g is None else (id(g), g._version)


class _SecondOrderStep(torch.autograd.Function):
    """out = p - lr * g(p).  Backward: v -> v - lr * H v, with H v supplied by `hvp` (finite differences of
    the first-order gradient at p +- e v; see the module docstring)."""

    @staticmethod
    def forward(ctx, p, g, lr, hvp, defer_to=None):
        ctx.lr, ctx.hvp, ctx.defer_to = lr, hvp, defer_to
        out = torch.empty_like(p)
        check(_lib.load().dyb_fastweight_update(p.data_ptr(), g.data_ptr(), out.data_ptr(), float(lr), p.numel(),
                                                stream_of(p)), "dyb_fastweight_update")
        return out

    @staticmethod
    def backward(ctx, d_out):
        # v_k = v_{k+1} - lr * H v_{k+1}: the same fused streaming launch as the fast-weight step (out = p - lr * g)
        d_out = d_out.contiguous()
        hv = ctx.hvp(d_out).contiguous()
        if ctx.defer_to is not None:
            # first inner step: its output gradient IS the parameter's.  Leave the accumulation to the optimiser's fused
            # "Adam + accumulate" launch (dynaboa_amd/optim.py): the parameter receives v and carries (H v, lr) beside it
            ctx.defer_to._so_pending = (hv, ctx.lr)
            return d_out, None, None, None, None
        out = torch.empty_like(d_out)
        check(_lib.load().dyb_fastweight_update(d_out.data_ptr(), hv.data_ptr(), out.data_ptr(), float(ctx.lr), d_out.numel(),
                                                stream_of(d_out)), "dyb_fastweight_update")
        return out, None, None, None, None


class MAML(nn.Module):
    # finite-difference step of the second-order path, relative to |theta| / |v|.  Swept on MI355X against the reference's
    # second-order goldens (tools/so_fd_probe.py, profiles/r02_so_fd_sweep.txt): per-tensor outer-gradient norm error
    # median / max at inner_step 2 and 3 = 8.6e-4 / 1.3e-2 and 7.5e-4 / 1.8e-2 at 1e-6 (fp32 rounding of the difference),
    # 3.7e-4 / 5.0e-3 and 1.2e-3 / 7.9e-3 at 1e-5, 1e-2 / 3.5e-2 and 3.4e-2 / 6.7e-2 at 1e-4, and garbage from 1e-3 on (the
    # perturbed forward leaves the linear region of thousands of ReLUs: this network is piecewise linear in theta, a large
    # step measures kink crossings, not curvature).  The first-order gradient is 19-24 % away from the second-order one.
    fd_rel = 1e-5

    def __init__(self, model: nn.Module, lr: float, first_order: bool = True, _theta=None):
        super().__init__()
        self.module = model
        self.lr = lr
        self.first_order = first_order
        self._theta = _theta            # None: this is the base wrapper; tensor: a learner's fast weights
        # second order with dynaboa_amd.optim.Adam: leave the last accumulation (v - lr * H v) to the optimiser's fused launch
        # (BaseAdaptor switches it on; off for stand-alone use, where any optimiser may read .grad)
        self.defer_accumulate = False
        self._first = True

    def forward(self, *args, **kwargs):
        if self._theta is None:
            return self.module(*args, **kwargs)
        return self.module(*args, theta=self._theta, **kwargs)

    def clone(self, first_order=None):
        """New learner whose weights are graph-connected to this one's (identity edge).  learn2learn
        copies every tensor (216 MB of traffic here); no caller mutates a learner in place, so the
        fast weights start as an autograd alias of theta and the first adapt() produces the copy."""
        fo = self.first_order if first_order is None else first_order
        src = self.module.theta if self._theta is None else self._theta
        learner = MAML(self.module, self.lr, fo, _theta=src.view_as(src))
        learner.defer_accumulate = self.defer_accumulate and self._theta is None      # only a clone of the base wrapper
        learner.train(self.training)
        return learner

    def adapt(self, loss, first_order=None, closure=None, hvp_factory=None):
        """hvp_factory (second order only): theta_k -> (v -> H v), an exact Hessian-vector product of this level's loss at
        the fast weights theta_k (dynaboa_amd/hvp.py); without it H v is the central difference of `closure`'s gradient."""
        fo = self.first_order if first_order is None else first_order
        if self._theta is None:
            raise RuntimeError("adapt() must be called on a clone()")
        (g,) = torch.autograd.grad(loss, [self._theta])
        if fo:
            self._theta = _FastWeightStep.apply(self._theta, g, self.lr)
            return
        # the first adapt() of a clone feeds the base parameter directly: its accumulation can ride in Adam's launch
        defer = self.module.theta if (self.defer_accumulate and self._first) else None
        self._first = False
        if defer is not None and not hasattr(defer, "_so_grad_hook"):
            defer._so_grad_hook = defer.register_post_accumulate_grad_hook(_note_grad)
        if hvp_factory is not None:
            exact = hvp_factory(self._theta.detach())
            self._theta = _SecondOrderStep.apply(self._theta, g, self.lr, lambda v: exact(v.detach()), defer)
            return
        if closure is None:
            raise NotImplementedError(
                "second-order adapt() needs the lower-level loss as a function of the learner: "
                "learner.adapt(loss, closure=lambda learner: <same loss, evaluated with `learner`>) - see dynaboa_amd/maml.py")
        theta_k = self._theta.detach()
        module, lr, training, fd_rel = self.module, self.lr, self.training, self.fd_rel

        def grad_at(theta):
            theta = theta.requires_grad_(True)
            probe = MAML(module, lr, True, _theta=theta)
            probe.train(training)
            with torch.enable_grad():
                (gs,) = torch.autograd.grad(closure(probe), [theta])
            return gs

        def hvp(v):
            v = v.detach()
            eps = fd_rel * torch.linalg.vector_norm(theta_k) / torch.linalg.vector_norm(v).clamp_min(1e-30)
            step = eps * v
            return (grad_at(theta_k + step) - grad_at(theta_k - step)) / (2 * eps)

        self._theta = _SecondOrderStep.apply(self._theta, g, self.lr, hvp, defer)

    def parameters(self, recurse: bool = True):
        if self._theta is None:
            return self.module.parameters()
        return iter([self._theta])

    # the reference checkpoint was saved from the wrapper: keys carry "module." (base_adaptor.py:116-125)
    def state_dict(self, *args, prefix="", **kw):
        sd = self.module.state_dict()
        return OrderedDict((prefix + "module." + k, v) for k, v in sd.items())

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        bad = [k for k in state_dict if not k.startswith("module.")]
        if strict and bad:
            raise RuntimeError(f"unexpected keys without 'module.' prefix: {bad[:5]}")
        return self.module.load_state_dict({k[len("module."):]: v for k, v in state_dict.items()
                                            if k.startswith("module.")}, strict=strict)
