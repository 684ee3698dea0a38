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
