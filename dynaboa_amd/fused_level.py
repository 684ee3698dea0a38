"""One autograd node per adaptation level: HMR forward -> SMPL (LBS) -> frame-loss head, and their
backward, as five C calls with nothing in between.

The reference builds the same thing from separate modules (``lower_level_adaptation`` /
``upper_level_adaptation``, reference base_adaptor.py:222-317: ``model(image)`` ->
``decode_smpl_params`` -> projection + masked 2-D MSE + shape prior + pose prior).  Composing
``HMR.forward``, ``SMPL.forward`` and ``losses.frame_losses`` gives the same numbers through three
autograd nodes plus ~20 small torch launches of glue per level (slices, ``cat``, ``ones_like``,
gradient scaling / accumulation); on a chain that is launch- and host-issue-bound that
glue is ~1 ms of host time and ~60 us of GPU time per level.  Here the outputs stay differentiable -
the teacher / motion / label terms of the full loss set attach to ``rotmat / shape / cam / joints``
exactly as before; their gradients arrive as the optional ``*_ext`` inputs of the combine kernel.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib, constants as C
from ._abi import check
from .hmr import (STATE_LD, _feature_views, aux_stream_of, get_bwd_stage, get_layout, get_workspace, stream_of)

_STAGE: Dict[tuple, dict] = {}
# The activation arena of the newest level forward per plan, for the exact Hessian-vector product of that level (dynaboa_amd/hvp.py),
# which starts from it instead of repeating the forward.  Identity of "the same forward" is NOT inferred from addresses alone: an
# entry keeps its three input tensors alive (so no later tensor can occupy their storage while the entry exists), compares the
# storages and autograd version counters (the wrappers around the library's in-place kernels - Adam, EMA, preprocess - bump them
# by hand), is handed out ONCE (the consumer pops it) and is dropped at the end of the frame and whenever a level runs unfused.
_LAST_FORWARD: Dict[tuple, tuple] = {}


def _fwd_key(theta, image, init_state, n_iter):
    return (theta.data_ptr(), theta._version, tuple(theta.shape), image.data_ptr(), image._version, tuple(image.shape),
            init_state.data_ptr(), init_state._version, int(n_iter))


def last_forward_acts(theta, image, init_state, n_iter):
    """The activation arena `level_forward` filled for exactly these (weights, image, initial state) if it was the newest one, else
    None; the entry is consumed either way."""
    B, _, H, W = image.shape
    hit = _LAST_FORWARD.pop((B, H, W, str(theta.device)), None)
    if hit is None or hit[0] != _fwd_key(theta, image, init_state, n_iter):
        return None
    return hit[1]


def clear_last_forward():
    """Drop every remembered level forward (end of a frame; a level that ran without the fused node)."""
    _LAST_FORWARD.clear()


def _stage(B: int, device: torch.device) -> dict:
    """Persistent scratch of the backward half, per (batch, device, stream)."""
    sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (B, str(device), sid)
    st = _STAGE.get(key)
    if st is None:
        lib = _lib.load()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        wsb = int(lib.dyb_lbs_bwd_workspace_bytes(B))
        st = dict(djoints=f(B, C.NUM_OUT_JOINTS, 3), drot_s=f(B, 24, 3, 3), dbetas_s=f(B, 10),
                  lbs_ws=torch.empty(wsb, dtype=torch.uint8, device=device), lbs_wsb=wsb)
        _STAGE[key] = st
    return st


def _c(t):
    return None if t is None else t.contiguous().float()


def _p(t):
    return None if t is None else t.data_ptr()


class _LevelFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, image, init_state, kp2d, smpl, prior, w2d, wshape, wpose, n_iter, need_feature):
        lib = _lib.load()
        B, _, H, W = image.shape
        L = get_layout(B, H, W)
        if theta.numel() != L.n_params:
            raise ValueError("parameter arena size does not match the engine plan")
        dev = theta.device
        st = stream_of(theta)
        image = image.contiguous().float()
        init_state = init_state.contiguous().float()
        kp2d = kp2d.contiguous().float()
        acts = torch.empty(L.act_floats, dtype=torch.float32, device=dev)
        ws = get_workspace(L, dev)
        check(lib.dyb_hmr_forward(L.plan, theta.data_ptr(), image.data_ptr(), init_state.data_ptr(), n_iter, acts.data_ptr(),
                                  ws.data_ptr(), L.ws_bytes, st), "dyb_hmr_forward")
        rot = acts[L.off_rotmat:L.off_rotmat + B * 216].view(B, 24, 3, 3)
        state = acts[L.off_state:L.off_state + B * STATE_LD].view(B, STATE_LD)
        shape, cam = state[:, 144:154], state[:, 154:157]
        # one allocation for everything the head produces: verts | joints | saved | losses(4) | loss-side gradients | parts
        nv, nj, ns = B * C.NUM_VERTS * 3, B * C.NUM_OUT_JOINTS * 3, int(lib.dyb_lbs_saved_floats(B))
        sizes = [nv, nj, ns, 4, B * 216, B * 10, B * 3, nj, B * 4]
        buf = torch.empty(sum((s + 3) & ~3 for s in sizes), dtype=torch.float32, device=dev)
        parts, o = [], 0
        for s in sizes:
            parts.append(buf[o:o + s])
            o += (s + 3) & ~3
        verts, joints, saved, losses, drot_l, dshape_l, dcam_l, djoints_l, lws = parts
        check(lib.dyb_lbs_fwd(smpl._pf, smpl._pi, shape.data_ptr(), STATE_LD, rot.data_ptr(), verts.data_ptr(), joints.data_ptr(),
                              saved.data_ptr(), B, st), "dyb_lbs_fwd")
        check(lib.dyb_frame_losses(rot.data_ptr(), shape.data_ptr(), STATE_LD, cam.data_ptr(), STATE_LD, joints.data_ptr(),
                                   kp2d.data_ptr(), prior.means.data_ptr(), prior.precisions.data_ptr(),
                                   prior.log_nll_weights.data_ptr(), float(w2d), float(wshape), float(wpose), losses.data_ptr(),
                                   drot_l.data_ptr(), dshape_l.data_ptr(), 10, dcam_l.data_ptr(), 3, djoints_l.data_ptr(), B,
                                   lws.data_ptr(), B * 16, st), "dyb_frame_losses")
        ctx.L, ctx.n_iter, ctx.smpl, ctx.B = L, n_iter, smpl, B
        ctx.set_materialize_grads(False)            # unused outputs arrive as None in backward (handled there), not as zero-filled tensors
        _LAST_FORWARD[(B, H, W, str(dev))] = (_fwd_key(theta, image, init_state, n_iter), acts, theta, image, init_state)
        ctx.save_for_backward(theta, acts, buf)
        ctx.sizes = sizes
        comps = losses[:3].clone()
        feats = tuple(_feature_views(L, acts, n_iter)) if need_feature else ()
        ctx.mark_non_differentiable(comps, *feats)
        return (losses[3], comps, rot, shape, cam, joints.view(B, C.NUM_OUT_JOINTS, 3), verts.view(B, C.NUM_VERTS, 3)) + feats

    @staticmethod
    def backward(ctx, g_total, _g_comps, d_rot, d_shape, d_cam, d_joints, d_verts, *_unused):
        lib = _lib.load()
        theta, acts, buf = ctx.saved_tensors
        L, B, smpl = ctx.L, ctx.B, ctx.smpl
        dev = theta.device
        st = stream_of(theta)
        parts, o = [], 0
        for s in ctx.sizes:
            parts.append(buf[o:o + s])
            o += (s + 3) & ~3
        _verts, _joints, saved, _losses, drot_l, dshape_l, dcam_l, djoints_l, _lws = parts
        rot = acts[L.off_rotmat:L.off_rotmat + B * 216]
        sc = _stage(B, dev)
        eng = get_bwd_stage(L, dev)
        g = _c(g_total)
        d_rot, d_shape, d_cam, d_joints, d_verts = _c(d_rot), _c(d_shape), _c(d_cam), _c(d_joints), _c(d_verts)
        if g is None and d_joints is None:
            sc["djoints"].zero_()
        else:
            a = djoints_l if g is not None else torch.zeros_like(djoints_l)
            check(lib.dyb_scale_add(_p(g), a.data_ptr(), _p(d_joints), sc["djoints"].data_ptr(), a.numel(), st), "dyb_scale_add")
        check(lib.dyb_lbs_bwd(smpl._pf, smpl._pi, rot.data_ptr(), saved.data_ptr(), sc["djoints"].data_ptr(), _p(d_verts),
                              sc["drot_s"].data_ptr(), sc["dbetas_s"].data_ptr(), 10, B, sc["lbs_ws"].data_ptr(), sc["lbs_wsb"],
                              st), "dyb_lbs_bwd")
        if g is None:                       # the loss total itself was not used: only the external gradients count
            g = torch.zeros((), device=dev)
        check(lib.dyb_head_grad_combine(g.data_ptr(), drot_l.data_ptr(), sc["drot_s"].data_ptr(), _p(d_rot), dshape_l.data_ptr(),
                                        sc["dbetas_s"].data_ptr(), _p(d_shape), dcam_l.data_ptr(), _p(d_cam),
                                        eng["d_rot"].data_ptr(), eng["d_state"].data_ptr(), B, st), "dyb_head_grad_combine")
        ws = get_workspace(L, dev)
        check(lib.dyb_hmr_backward(L.plan, theta.data_ptr(), acts.data_ptr(), eng["d_rot"].data_ptr(), eng["d_state"].data_ptr(),
                                   ctx.n_iter, eng["grads"].data_ptr(), ws.data_ptr(), L.ws_bytes, st, aux_stream_of(theta)),
              "dyb_hmr_backward")
        return (eng["grads"].clone(),) + (None,) * 10


def level_forward(model, smpl, prior, image, kp2d, w2d, wshape, wpose, n_iter: int = 3, need_feature: bool = True):
    """``model``: an ``HMR`` or a ``MAML`` wrapper / learner around one.  Returns
    ``(loss_total, comps(s2d, shape_prior, pose_prior), rotmat, shape, cam, joints49, verts, features)`` -
    ``loss_total`` and the five outputs are differentiable w.r.t. the (fast) weights."""
    hmr = getattr(model, "module", model)
    theta = getattr(model, "_theta", None)
    if theta is None:
        theta = hmr.theta
    if hmr.training:
        raise NotImplementedError("the fused level node is the eval-mode path (the adaptation loop runs model.eval(), "
                                  "dynaboa_benchmark.py:89); in train() mode use fused_level=0 / the HMR module directly")
    st0 = hmr.make_init_state(image.shape[0])
    out = _LevelFunction.apply(theta, image, st0, kp2d, smpl, prior, w2d, wshape, wpose, n_iter, need_feature)
    return out[0], out[1], out[2], out[3], out[4], out[5], out[6], list(out[7:])
