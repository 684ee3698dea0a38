"""Exact Hessian-vector products of a frame-loss adaptation level (second-order MAML, ``--hvp exact``).

``MAML.adapt`` in second-order mode needs  H v  with H the Hessian of the lower-level loss at the current fast weights
(reference base_adaptor.py:119 with ``first_order=False``; learn2learn's ``create_graph=True``).  The default
(dynaboa_amd/maml.py) takes it as a central difference of two first-order gradients of the whole level; here it is formed
forward-over-reverse through the network by the library's tangent passes (csrc/hvp_engine.inc, csrc/hvp_kernels.hip):

    tangent pass        d/de activations(theta + e v)                         dyb_hmr_jvp_forward
    head                H_head . tstate  (rot6d -> SMPL -> projection / priors on 157 numbers per sample)
    tangent of backward d/de grad_theta                                       dyb_hmr_jvp_backward  = H v

The head's second derivative is taken as a central difference of ITS analytic gradient along the state tangent - a smooth
function of 157 inputs per sample evaluated by the same loss / LBS kernels, where a difference quotient is accurate to ~1e-5 -
while the 50-layer ReLU / GroupNorm backbone, where a difference quotient of the whole network is noisy element-wise, is exact.

Covers levels made of the frame losses only (2-D keypoints + shape prior + pose prior): the benchmarked second-order
configuration.  Levels with teacher / motion / labelled terms fall back to the difference quotient (``frame_level_hvp``
returns None)."""
from __future__ import annotations

import torch

from . import _lib, constants as C
from ._abi import check
from .hmr import STATE_LD, get_layout, get_workspace, stream_of

HEAD_FD_REL = 2e-3          # |e * tstate| / |state| of the head's central difference


def _head_grad(lib, smpl, prior, state, kp2d, w2d, wshape, wpose, st):
    """Gradient of the frame-loss head w.r.t. the regressor state [B][160] (pose6d | shape | cam | pad), rot6d included.
    Same five C calls as fused_level._LevelFunction (forward head + its backward), on a given state."""
    B = state.shape[0]
    dev = state.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    rot = f(B, 24, 3, 3)
    check(lib.dyb_rot6d_fwd(state.data_ptr(), STATE_LD, rot.data_ptr(), B, st), "dyb_rot6d_fwd")
    shape_p, cam_p = state.data_ptr() + 144 * 4, state.data_ptr() + 154 * 4
    verts, joints = f(B, C.NUM_VERTS, 3), f(B, C.NUM_OUT_JOINTS, 3)
    saved = f(int(lib.dyb_lbs_saved_floats(B)))
    check(lib.dyb_lbs_fwd(smpl._pf, smpl._pi, shape_p, STATE_LD, rot.data_ptr(), verts.data_ptr(), joints.data_ptr(), saved.data_ptr(),
                          B, st), "dyb_lbs_fwd")
    losses, drot_l, dshape_l, dcam_l = f(4), f(B, 216), f(B, 10), f(B, 3)
    djoints_l, lws = f(B, C.NUM_OUT_JOINTS, 3), f(B * 16)
    check(lib.dyb_frame_losses(rot.data_ptr(), shape_p, STATE_LD, cam_p, STATE_LD, joints.data_ptr(), kp2d.data_ptr(),
                               prior.means.data_ptr(), prior.precisions.data_ptr(), prior.log_nll_weights.data_ptr(), float(w2d),
                               float(wshape), float(wpose), losses.data_ptr(), drot_l.data_ptr(), dshape_l.data_ptr(), 10,
                               dcam_l.data_ptr(), 3, djoints_l.data_ptr(), B, lws.data_ptr(), B * 16, st), "dyb_frame_losses")
    drot_s, dbetas_s = f(B, 216), f(B, 10)
    wsb = int(lib.dyb_lbs_bwd_workspace_bytes(B))
    lbs_ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(lib.dyb_lbs_bwd(smpl._pf, smpl._pi, rot.data_ptr(), saved.data_ptr(), djoints_l.data_ptr(), None, drot_s.data_ptr(),
                          dbetas_s.data_ptr(), 10, B, lbs_ws.data_ptr(), wsb, st), "dyb_lbs_bwd")
    d_rot = f(B, 216)
    d_state = torch.zeros(B, STATE_LD, dtype=torch.float32, device=dev)
    check(lib.dyb_head_grad_combine(None, drot_l.data_ptr(), drot_s.data_ptr(), None, dshape_l.data_ptr(), dbetas_s.data_ptr(), None,
                                    dcam_l.data_ptr(), None, d_rot.data_ptr(), d_state.data_ptr(), B, st), "dyb_head_grad_combine")
    check(lib.dyb_rot6d_bwd(state.data_ptr(), STATE_LD, d_rot.data_ptr(), d_state.data_ptr(), STATE_LD, B, st), "dyb_rot6d_bwd")
    return d_state


def frame_level_hvp(hmr, smpl, prior, theta, image, kp2d, w2d, wshape, wpose, n_iter: int = 3):
    """-> callable v -> H v for L(theta) = the frame-loss level (fused_level.level_forward's loss_total) at `theta`
    (a parameter arena: the learner's current fast weights), on (image, kp2d)."""
    lib = _lib.load()
    B, _, H, W = image.shape
    L = get_layout(B, H, W)
    theta = theta.detach()
    image = image.contiguous().float()
    kp2d = kp2d.contiguous().float()
    init_state = hmr.make_init_state(B).contiguous().float()
    if hmr.training:
        raise NotImplementedError("exact Hessian-vector products are the eval-mode path")

    def hvp(v):
        v = v.detach().contiguous().float()
        dev = theta.device
        st = stream_of(theta)
        ws = get_workspace(L, dev)
        acts = torch.empty(L.act_floats, dtype=torch.float32, device=dev)
        check(lib.dyb_hmr_forward(L.plan, theta.data_ptr(), image.data_ptr(), init_state.data_ptr(), n_iter, acts.data_ptr(),
                                  ws.data_ptr(), L.ws_bytes, st), "dyb_hmr_forward")
        dual = torch.empty(int(lib.dyb_hmr_hvp_dual_floats(L.plan)), dtype=torch.float32, device=dev)
        check(lib.dyb_hmr_jvp_forward(L.plan, theta.data_ptr(), v.data_ptr(), acts.data_ptr(), dual.data_ptr(), n_iter, ws.data_ptr(),
                                      L.ws_bytes, st), "dyb_hmr_jvp_forward")
        state = acts[L.off_state:L.off_state + B * STATE_LD].view(B, STATE_LD)
        off = int(lib.dyb_hmr_hvp_offset_tstate(L.plan))
        tstate = dual[off:off + B * STATE_LD].view(B, STATE_LD)
        g0 = _head_grad(lib, smpl, prior, state.contiguous(), kp2d, w2d, wshape, wpose, st)
        tn = torch.linalg.vector_norm(tstate[:, :157])
        eps = HEAD_FD_REL * torch.linalg.vector_norm(state[:, :157]) / tn.clamp_min(1e-30)
        step = torch.zeros_like(state)
        step[:, :157] = eps * tstate[:, :157]
        gp = _head_grad(lib, smpl, prior, (state + step).contiguous(), kp2d, w2d, wshape, wpose, st)
        gm = _head_grad(lib, smpl, prior, (state - step).contiguous(), kp2d, w2d, wshape, wpose, st)
        td = ((gp - gm) / (2 * eps)).contiguous()
        hv = torch.zeros(L.n_params, dtype=torch.float32, device=dev)
        check(lib.dyb_hmr_jvp_backward(L.plan, theta.data_ptr(), v.data_ptr(), acts.data_ptr(), dual.data_ptr(), g0.data_ptr(),
                                       td.data_ptr(), n_iter, hv.data_ptr(), ws.data_ptr(), L.ws_bytes, st), "dyb_hmr_jvp_backward")
        return hv
    return hvp
