"""Differentiable loss pieces on the HIP kernels (csrc/losses.hip): the camera projection of
reference base_adaptor.py:160-170, the GMM pose prior (base_adaptor.py:405-409 +
utils/smplify/prior.py:181-196) and the fused frame-loss block of lower/upper_level_adaptation."""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, assets
from ._abi import check
from .hmr import stream_of


class MaxMixturePrior(nn.Module):
    """Buffers of the 8-Gaussian max-mixture pose prior (reference utils/smplify/prior.py:100-160).
    ``prior_folder`` may hold the original ``gmm_08.pkl``; default is the float32 re-export shipped
    in dynaboa_amd/assets/."""

    def __init__(self, prior_folder: Optional[str] = None, num_gaussians: int = 8, dtype=torch.float32, **kw):
        super().__init__()
        import os
        path = None
        if prior_folder is not None:
            cand = os.path.join(prior_folder, f"gmm_{num_gaussians:02d}.pkl")
            if os.path.exists(cand):
                path = cand
        buf = assets.load_gmm_prior(path)
        self.register_buffer("means", torch.from_numpy(buf["means"]).float().contiguous())
        self.register_buffer("precisions", torch.from_numpy(buf["precisions"]).float().contiguous())
        self.register_buffer("nll_weights", torch.from_numpy(buf["nll_weights"]).float().contiguous())
        with warnings.catch_warnings(), np.errstate(divide="ignore"):
            warnings.simplefilter("ignore")
            logw = np.log(buf["nll_weights"].astype(np.float32)).reshape(-1)   # -inf for underflowed weights, as torch.log gives
        self.register_buffer("log_nll_weights", torch.from_numpy(logw).contiguous())

    def forward(self, pose, betas=None):
        """merged_log_likelihood (reference utils/smplify/prior.py:181-196; the call ``self.gmm_f(body_pose, betas)`` of
        base_adaptor.py:405-409): pose (B, 69) axis-angle body pose -> (B,) min over the mixture, differentiable in the pose;
        ``betas`` is unused, as in the reference.  (The adaptation path itself uses ``frame_losses`` / ``pose_prior``, where the
        rotation-matrix -> axis-angle conversion and the mixture are one fused kernel.)"""
        return _GmmPrior.apply(pose.reshape(-1, 69), self)

    merged_log_likelihood = forward


class _GmmPrior(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, prior):
        pose = pose.contiguous().float()
        B = pose.shape[0]
        out, dpose = torch.empty(B, device=pose.device), torch.empty_like(pose)
        check(_lib.load().dyb_gmm_prior(pose.data_ptr(), prior.means.data_ptr(), prior.precisions.data_ptr(),
                                        prior.log_nll_weights.data_ptr(), out.data_ptr(), dpose.data_ptr(), B, stream_of(pose)), "dyb_gmm_prior")
        ctx.save_for_backward(dpose)
        return out

    @staticmethod
    def backward(ctx, g):
        (dpose,) = ctx.saved_tensors
        return dpose * g.reshape(-1, 1), None


class _FrameLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rotmat, shape, cam, joints, kp2d, prior, w2d, wshape, wpose):
        lib = _lib.load()
        B = rotmat.shape[0]
        dev = rotmat.device
        rows = lambda t: t.float() if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous().float()   # row stride is passed
        rotmat, shape, cam = rotmat.contiguous().float(), rows(shape), rows(cam)
        joints, kp2d = joints.contiguous().float(), kp2d.contiguous().float()
        losses = torch.empty(4, device=dev)
        drot = torch.empty(B, 24, 3, 3, device=dev)
        dshape = torch.empty(B, 10, device=dev)
        dcam = torch.empty(B, 3, device=dev)
        djoints = torch.empty(B, 49, 3, device=dev)
        ws = torch.empty(B * 4, device=dev)
        check(lib.dyb_frame_losses(rotmat.data_ptr(), shape.data_ptr(), shape.stride(0), cam.data_ptr(), cam.stride(0),
                                   joints.data_ptr(), kp2d.data_ptr(), prior.means.data_ptr(), prior.precisions.data_ptr(),
                                   prior.log_nll_weights.data_ptr(), float(w2d), float(wshape), float(wpose),
                                   losses.data_ptr(), drot.data_ptr(), dshape.data_ptr(), 10, dcam.data_ptr(), 3,
                                   djoints.data_ptr(), B, ws.data_ptr(), B * 16, stream_of(rotmat)), "dyb_frame_losses")
        ctx.save_for_backward(drot, dshape, dcam, djoints)
        comps = losses[:3].clone()
        ctx.mark_non_differentiable(comps)
        return losses[3], comps

    @staticmethod
    def backward(ctx, g_total, _g_comps):
        drot, dshape, dcam, djoints = ctx.saved_tensors
        return drot * g_total, dshape * g_total, dcam * g_total, djoints * g_total, None, None, None, None, None


def frame_losses(rotmat, shape, cam, joints49, kp2d, prior: MaxMixturePrior, w2d, wshape, wpose):
    """-> (weighted total [differentiable], tensor(s2dloss, shape_prior, pose_prior) [logging only])."""
    return _FrameLoss.apply(rotmat, shape, cam, joints49, kp2d, prior, w2d, wshape, wpose)


def pose_prior(rotmat, prior: MaxMixturePrior):
    """cal_pose_prior (reference base_adaptor.py:405-409) as a differentiable scalar."""
    B = rotmat.shape[0]
    z = torch.zeros(B, 10, device=rotmat.device)
    cam = torch.ones(B, 3, device=rotmat.device)
    j = torch.ones(B, 49, 3, device=rotmat.device)
    total, _ = _FrameLoss.apply(rotmat, z, cam, j, torch.zeros(B, 49, 3, device=rotmat.device), prior, 0.0, 0.0, 1.0)
    return total


class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam, p3):
        lib = _lib.load()
        B, n = p3.shape[0], p3.shape[1]
        cam, p3 = cam.contiguous().float(), p3.contiguous().float()
        p2 = torch.empty(B, n, 2, device=p3.device)
        check(lib.dyb_projection_fwd(cam.data_ptr(), cam.stride(0), p3.data_ptr(), p2.data_ptr(), B, n, stream_of(p3)),
              "dyb_projection_fwd")
        ctx.save_for_backward(cam, p3)
        return p2

    @staticmethod
    def backward(ctx, g2):
        lib = _lib.load()
        cam, p3 = ctx.saved_tensors
        B, n = p3.shape[0], p3.shape[1]
        g2 = g2.contiguous().float()
        dp3 = torch.empty_like(p3)
        dcam = torch.empty(B, 3, device=p3.device)
        check(lib.dyb_projection_bwd(cam.data_ptr(), cam.stride(0), p3.data_ptr(), g2.data_ptr(), dp3.data_ptr(),
                                     dcam.data_ptr(), 3, B, n, stream_of(p3)), "dyb_projection_bwd")
        return dcam, dp3


def projection_normed(cam, s3d):
    """[-1,1]-normalised weak-perspective projection, focal 5000, crop 224 (base_adaptor.py:160-170)."""
    return _Projection.apply(cam, s3d)


class _AuxTerms(torch.autograd.Function):
    """One of the three non-frame terms of a level (csrc/losses.hip dyb_aux_loss_terms: value + gradient in one launch) as an autograd
    node.  Differentiable in the student pass's (rot, shape, cam, joints) and, for the motion term, in the history pass's
    (cam2, joints2)."""

    @staticmethod
    def forward(ctx, mode, rot, shape, cam, joints, rot2, shape2, cam2, joints2, kp, kp2, gt_rot, gt_betas, gt_s3d):
        lib = _lib.load()
        B = rot.shape[0]
        dev = rot.device
        rows = lambda t: None if t is None else (t.float() if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous().float())
        full = lambda t: None if t is None else t.contiguous().float()
        ld = lambda t: 0 if t is None else t.stride(0)
        p = lambda t: None if t is None else t.data_ptr()
        rot, joints, rot2, joints2 = full(rot), full(joints), full(rot2), full(joints2)
        shape, cam, shape2, cam2 = rows(shape), rows(cam), rows(shape2), rows(cam2)
        kp, kp2, gt_rot, gt_betas, gt_s3d = full(kp), full(kp2), full(gt_rot), full(gt_betas), full(gt_s3d)
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        vals, d_rot, d_shape, d_cam, d_joints = f(5), f(B, 24, 3, 3), f(B, 10), f(B, 3), f(B, 49, 3)
        d_cam2, d_joints2 = (f(B, 3), f(B, 49, 3)) if mode == 1 else (None, None)
        check(lib.dyb_aux_loss_terms(mode, B, 0, 1.0, rot.data_ptr(), shape.data_ptr(), ld(shape), cam.data_ptr(), ld(cam), joints.data_ptr(),
                                     p(rot2), p(shape2), ld(shape2), p(cam2), ld(cam2), p(joints2), p(kp), p(kp2), p(gt_rot), p(gt_betas),
                                     p(gt_s3d), vals.data_ptr(), d_rot.data_ptr(), d_shape.data_ptr(), d_cam.data_ptr(),
                                     d_joints.data_ptr(), p(d_cam2), p(d_joints2), stream_of(rot)), "dyb_aux_loss_terms")
        ctx.mode = mode
        ctx.save_for_backward(*(t for t in (d_rot, d_shape, d_cam, d_joints, d_cam2, d_joints2) if t is not None))
        comps = vals[:4].clone()
        ctx.mark_non_differentiable(comps)
        return vals[4], comps

    @staticmethod
    def backward(ctx, g, _g_comps):
        saved = ctx.saved_tensors
        d_rot, d_shape, d_cam, d_joints = saved[:4]
        out = [None, d_rot * g, d_shape * g, d_cam * g, d_joints * g, None, None, None, None]
        if ctx.mode == 1:
            out[7], out[8] = saved[4] * g, saved[5] * g
        return tuple(out) + (None,) * 5


AUX_MAX_BATCH = 16          # dyb_aux_loss_terms: one launch covers up to 16 samples


def _aux(mode, batched, const):
    """`batched`: 13 per-sample tensors (or None) in _AuxTerms.forward's order.  Every component is a mean over the batch, so a batch
    beyond one launch's 16 samples is the size-weighted sum of its chunks."""
    B = batched[0].shape[0]
    if B <= AUX_MAX_BATCH:
        return _AuxTerms.apply(mode, *batched)
    loss, comps = None, None
    for i in range(0, B, AUX_MAX_BATCH):
        part = [None if t is None else t[i:i + AUX_MAX_BATCH] for t in batched]
        l, c = _AuxTerms.apply(mode, *part)
        wgt = part[0].shape[0] / B
        loss = l * wgt if loss is None else loss + l * wgt
        comps = c * wgt if comps is None else comps + c * wgt
    return loss, comps


def teacher_term(rot, shape, cam, joints49, t_rot, t_shape, t_cam, t_joints49):
    """Mean-teacher consistency (reference base_adaptor.py:320-343): 5 mse(s2d) + 5 mse(s3d) + 0.001 mse(shape) + mse(rotmat) against
    the teacher's outputs, the normalised projections formed inside -> (loss, tensor(s2d, s3d, shape, pose) for logging)."""
    return _aux(0, [rot, shape, cam, joints49, t_rot, t_shape, t_cam, t_joints49, None, None, None, None, None], None)


def motion_term(rot, shape, cam, joints49, h_cam, h_joints49, kp2d, hist_kp2d):
    """Motion term (base_adaptor.py:379-398): confidence-masked mse between the predicted and the annotated keypoint motion from the
    history frame to this one; differentiable in both passes -> (loss, components)."""
    return _aux(1, [rot, shape, cam, joints49, None, None, h_cam, h_joints49, kp2d, hist_kp2d, None, None, None], None)


def labelled_term(rot, shape, cam, joints49, kp2d, gt_rot, gt_betas, gt_s3d):
    """Labelled-exemplar term (base_adaptor.py:346-376 with the hip-centred 3-D loss of :412-422) -> (loss, tensor(s2d, s3d, shape,
    pose))."""
    return _aux(2, [rot, shape, cam, joints49, None, None, None, None, kp2d, None, gt_rot, gt_betas, gt_s3d], None)
