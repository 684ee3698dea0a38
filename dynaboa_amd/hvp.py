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

``frame_level_hvp`` covers levels made of the frame losses only (2-D keypoints + shape prior + pose prior) - the benchmarked
second-order configuration, checked on the GPU against the reference's second-order golden.  ``general_level_hvp`` covers any
level (teacher / motion / labelled terms: up to three network passes sharing the weights, head assembled from the adaptor's
differentiable pieces); checked on MI355X against the reference's own second-order run on its default term set (golden
g5_so_inner1_full, 4 frames, element-wise slices of Adam's moments included) and against the oracle's create_graph gradient on the
CPU emulator (1e-4 at layer2, 1e-6 above, where the difference quotient gave 2e-2 / 6e-3); the default (``--hvp_terms all``)."""
from __future__ import annotations

import torch

from . import _lib, constants as C
from ._abi import check
from .hmr import STATE_LD, aux_stream_of, get_layout, get_workspace, stream_of

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
    from .fused_level import last_forward_acts
    # the level was just evaluated at these weights: its activations are the primal pass (nothing writes to them afterwards)
    level_acts = last_forward_acts(theta, image, init_state, n_iter)
    kp3 = kp2d.repeat(3, 1, 1)

    def hvp(v):
        v = v.detach().contiguous().float()
        dev = theta.device
        st = stream_of(theta)
        ws = get_workspace(L, dev)
        acts = level_acts
        if acts is None:
            acts = torch.empty(L.act_floats, dtype=torch.float32, device=dev)
            check(lib.dyb_hmr_forward(L.plan, theta.data_ptr(), image.data_ptr(), init_state.data_ptr(), n_iter, acts.data_ptr(),
                                      ws.data_ptr(), L.ws_bytes, st), "dyb_hmr_forward")
        dual = torch.empty(int(lib.dyb_hmr_hvp_dual_floats(L.plan)), dtype=torch.float32, device=dev)
        check(lib.dyb_hmr_jvp_forward(L.plan, theta.data_ptr(), v.data_ptr(), acts.data_ptr(), dual.data_ptr(), n_iter, ws.data_ptr(),
                                      L.ws_bytes, st, aux_stream_of(theta)), "dyb_hmr_jvp_forward")
        state = acts[L.off_state:L.off_state + B * STATE_LD].view(B, STATE_LD)
        off = int(lib.dyb_hmr_hvp_offset_tstate(L.plan))
        tstate = dual[off:off + B * STATE_LD].view(B, STATE_LD)
        tn = torch.linalg.vector_norm(tstate[:, :157])
        eps = HEAD_FD_REL * torch.linalg.vector_norm(state[:, :157]) / tn.clamp_min(1e-30)
        step = torch.zeros_like(state)
        step[:, :157] = eps * tstate[:, :157]
        # the head's gradient at the state and at the two difference points as ONE batch of 3 B samples (a third of the launches):
        # every term of the head is a mean over the batch, so each sample's gradient comes out scaled by 1 / 3
        g3 = _head_grad(lib, smpl, prior, torch.cat([state, state + step, state - step]).contiguous(), kp3, w2d, wshape, wpose, st) * 3.0
        g0, gp, gm = g3[:B].contiguous(), g3[B:2 * B], g3[2 * B:]
        td = ((gp - gm) / (2 * eps)).contiguous()
        hv = torch.zeros(L.n_params, dtype=torch.float32, device=dev)
        check(lib.dyb_hmr_jvp_backward(L.plan, theta.data_ptr(), v.data_ptr(), acts.data_ptr(), dual.data_ptr(), g0.data_ptr(),
                                       td.data_ptr(), n_iter, hv.data_ptr(), ws.data_ptr(), L.ws_bytes, st, aux_stream_of(theta)),
              "dyb_hmr_jvp_backward")
        return hv
    return hvp


# ---------------------------------------------------------------------------------------------------------------------
# Any level: frame losses + mean-teacher + motion + labelled-exemplar terms (reference base_adaptor.py:222-398)
# ---------------------------------------------------------------------------------------------------------------------
def rot6d_to_rotmat(x6: torch.Tensor) -> torch.Tensor:
    """reference utils/geometry.py:47-61 on the regressor's 144 pose numbers -> (B, 24, 3, 3), on the library's kernel pair
    (geometry._Rot6d: forward + hand-derived backward)."""
    from .geometry import rot6d_to_rotmat as r6
    return r6(x6.reshape(-1, 144)).view(-1, 24, 3, 3)


class _Pass:
    """One network pass of a level (an image batch evaluated with the level's weights): primal activations, tangent arena."""

    def __init__(self, lib, hmr, theta, image, n_iter, acts=None):
        self.lib, self.n_iter = lib, n_iter
        self.image = image.contiguous().float()
        B, _, H, W = self.image.shape
        self.B, self.L = B, get_layout(B, H, W)
        L, dev = self.L, theta.device
        self.ws = get_workspace(L, dev)
        self.init_state = hmr.make_init_state(B).contiguous().float()
        self.acts = acts                                 # the level's own forward of this image, when the caller still has it
        if self.acts is None:
            self.acts = torch.empty(L.act_floats, dtype=torch.float32, device=dev)
            check(lib.dyb_hmr_forward(L.plan, theta.data_ptr(), self.image.data_ptr(), self.init_state.data_ptr(), n_iter,
                                      self.acts.data_ptr(), self.ws.data_ptr(), L.ws_bytes, stream_of(theta)), "dyb_hmr_forward")
        self.state = self.acts[L.off_state:L.off_state + B * STATE_LD].view(B, STATE_LD).clone()
        self.dual = None

    def tangent(self, theta, v):
        L, lib = self.L, self.lib
        self.dual = torch.empty(int(lib.dyb_hmr_hvp_dual_floats(L.plan)), dtype=torch.float32, device=theta.device)
        check(lib.dyb_hmr_jvp_forward(L.plan, theta.data_ptr(), v.data_ptr(), self.acts.data_ptr(), self.dual.data_ptr(), self.n_iter,
                                      self.ws.data_ptr(), L.ws_bytes, stream_of(theta), aux_stream_of(theta)), "dyb_hmr_jvp_forward")
        off = int(lib.dyb_hmr_hvp_offset_tstate(L.plan))
        return self.dual[off:off + self.B * STATE_LD].view(self.B, STATE_LD).clone()

    def hv(self, theta, v, g, tg):
        L, lib = self.L, self.lib
        out = torch.zeros(L.n_params, dtype=torch.float32, device=theta.device)
        g, tg = g.contiguous().float(), tg.contiguous().float()
        check(lib.dyb_hmr_jvp_backward(L.plan, theta.data_ptr(), v.data_ptr(), self.acts.data_ptr(), self.dual.data_ptr(), g.data_ptr(),
                                       tg.data_ptr(), self.n_iter, out.data_ptr(), self.ws.data_ptr(), L.ws_bytes, stream_of(theta),
                                       aux_stream_of(theta)), "dyb_hmr_jvp_backward")
        return out


def general_level_hvp(ad, level, hmr, theta, image, kp2d, h36m_batch, n_iter: int = 3):
    """-> callable v -> H v for ANY level `BaseAdaptor._level` can assemble: the level's loss is a function of the regressor
    states of up to three network passes that share the weights - the frame (frame losses, teacher term, motion term), the
    frame `interval` steps back (motion term) and the retrieved exemplar images (labelled term).  Per pass a tangent pass
    gives the state tangent; the head - rot6d, SMPL, projection and the loss terms, assembled here exactly as `_level` does,
    from the same differentiable pieces - is differentiated by torch for its gradient and by a central difference of that
    gradient along the joint state tangent for its second derivative; per pass the tangent of the backward gives its share of
    H v.  Exemplars, teacher targets and history are those of the level evaluation (nothing is re-drawn)."""
    from .losses import frame_losses, labelled_term, motion_term, teacher_term
    lib = _lib.load()
    o = ad.options
    theta = theta.detach()
    if hmr.training:
        raise NotImplementedError("exact Hessian-vector products are the eval-mode path")
    use_frame = bool(getattr(o, f"use_frame_losses_{level}"))
    temporal = bool(getattr(o, f"use_temporal_losses_{level}"))
    use_teacher = temporal and bool(o.use_meanteacher)
    use_motion = temporal and bool(o.use_motion) and (ad.global_step - o.interval) > 0
    use_label = bool(getattr(o, f"{level}_level_mixtrain"))
    kp2d = kp2d.contiguous().float()
    teacher_t = None
    if use_teacher:
        with torch.no_grad():
            t_rot, t_shape, t_cam = ad.teacher(image)
            t_s3d = ad.decode_smpl_params(t_rot, t_shape)["s3d"]
            teacher_t = (t_rot, t_shape, t_cam, t_s3d)
    hist = ad.get_hist() if use_motion else None
    from .fused_level import last_forward_acts
    # the frame's forward at these weights was just evaluated by the level: its activations are the primal pass of "img"
    img_acts = last_forward_acts(theta, image.contiguous().float(), hmr.make_init_state(image.shape[0]).contiguous().float(), n_iter)

    def preds(state):
        rot = rot6d_to_rotmat(state[:, :144])
        shape, cam = state[:, 144:154], state[:, 154:157]
        return rot, shape, cam, ad.decode_smpl_params(rot, shape)["s3d"]

    rep3 = lambda t: t.repeat((3,) + (1,) * (t.dim() - 1))
    consts = {}

    def const(name, make):                             # the level's constants, tiled over the three evaluation points once
        if name not in consts:
            consts[name] = make()
        return consts[name]

    def head(states):
        """states: dict pass name -> (3 B, 160): the pass's state and the two difference points state +- eps * tstate as ONE batch
        (a third of the launches).  Every term is a mean over the batch, so the value is the mean of the three losses and each
        point's gradient comes out scaled by 1 / 3."""
        rot, shape, cam, s3d = preds(states["img"])
        kp3 = const("kp", lambda: rep3(kp2d))
        loss = None
        if use_frame:
            loss = frame_losses(rot, shape, cam, s3d, kp3, ad.gmm_f, o.s2dloss_weight, o.shape_prior_weight, o.pose_prior_weight)[0]
        # teacher / motion / labelled terms: the value+gradient kernel nodes `_level` uses (losses._AuxTerms)
        if temporal:
            if use_teacher:                             # base_adaptor.py:320-343
                t_rot, t_shape, t_cam, t_s3d = const("teacher", lambda: tuple(rep3(t) for t in teacher_t))
                t = teacher_term(rot, shape, cam, s3d, t_rot, t_shape, t_cam, t_s3d)[0] * o.teacherloss_weight
                loss = t if loss is None else loss + t
            if use_motion:                              # base_adaptor.py:379-398
                h_rot, h_shape, h_cam, h_s3d = preds(states["hist"])
                loss = loss + motion_term(rot, shape, cam, s3d, h_cam, h_s3d, kp3, const("hist_kp", lambda: rep3(hist[1])))[0] * o.motionloss_weight
        if use_label:                                   # base_adaptor.py:346-376
            from .geometry import batch_rodrigues
            b = h36m_batch
            e_rot, e_shape, e_cam, e_s3d = preds(states["ex"])
            gt_rot, e_kp, e_betas, e_p3 = const("ex", lambda: (rep3(batch_rodrigues(b["pose"].view(-1, 3)).view(-1, 24, 3, 3)),
                                                               rep3(b["keypoints"]), rep3(b["betas"]), rep3(b["pose_3d"])))
            lab = labelled_term(e_rot, e_shape, e_cam, e_s3d, e_kp, gt_rot, e_betas, e_p3)[0]
            loss = loss + lab * o.labelloss_weight
        return loss

    def head_grad(states):
        with torch.enable_grad():
            leaf = {k: s.detach().clone().requires_grad_(True) for k, s in states.items()}
            gs = torch.autograd.grad(head(leaf), list(leaf.values()), allow_unused=True)
        return {k: (torch.zeros_like(s) if g is None else g) for (k, s), g in zip(leaf.items(), gs)}

    def hvp(v):
        v = v.detach().contiguous().float()
        with torch.no_grad():
            passes = {"img": _Pass(lib, hmr, theta, image, n_iter, acts=img_acts)}
            if use_motion:
                passes["hist"] = _Pass(lib, hmr, theta, hist[0], n_iter)
            if use_label:
                passes["ex"] = _Pass(lib, hmr, theta, h36m_batch["img"], n_iter)
            states = {k: p.state for k, p in passes.items()}
            tst = {k: p.tangent(theta, v) for k, p in passes.items()}
            for t in tst.values():
                t[:, 157:] = 0
            sn = torch.sqrt(sum((s[:, :157] ** 2).sum() for s in states.values()))
            tn = torch.sqrt(sum((t ** 2).sum() for t in tst.values()))
            eps = HEAD_FD_REL * sn / tn.clamp_min(1e-30)
        g3 = head_grad({k: torch.cat([states[k], states[k] + eps * tst[k], states[k] - eps * tst[k]]) for k in states})
        with torch.no_grad():
            out = None
            for k, p in passes.items():
                g0, gp, gm = (3.0 * g3[k]).chunk(3)
                h = p.hv(theta, v, g0, (gp - gm) / (2 * eps))
                out = h if out is None else out + h
        return out
    return hvp
