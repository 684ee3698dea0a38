"""Rotation conversions and the camera projection with the reference's names (``utils/geometry.py``): rot6d_to_rotmat (:47-61),
rotation_matrix_to_angle_axis (:184-213), perspective_projection (:63-91) on HIP kernels with hand-derived backward;
batch_rodrigues (:9-24) on the rodrigues kernel when no gradient flows (it only converts the ground-truth pose of retrieved
exemplars / metric targets, never a learned quantity), as a few torch ops otherwise."""
from __future__ import annotations

import torch

from . import _lib
from ._abi import check
from .hmr import stream_of


class _Rot6d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        B = x.shape[0]
        R = torch.empty(B * 24, 3, 3, device=x.device)
        check(_lib.load().dyb_rot6d_fwd(x.data_ptr(), x.stride(0), R.data_ptr(), B, stream_of(x)), "dyb_rot6d_fwd")
        ctx.save_for_backward(x)
        return R

    @staticmethod
    def backward(ctx, dR):
        (x,) = ctx.saved_tensors
        B = x.shape[0]
        dx = torch.empty_like(x)
        dR = dR.contiguous().float()
        check(_lib.load().dyb_rot6d_bwd(x.data_ptr(), x.stride(0), dR.data_ptr(), dx.data_ptr(), dx.stride(0), B,
                                        stream_of(x)), "dyb_rot6d_bwd")
        return dx


def rot6d_to_rotmat(x):
    """(B,144) -> (B*24,3,3)."""
    return _Rot6d.apply(x.reshape(-1, 144))


class _R2AA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R):
        R = R.contiguous().float()
        n = R.shape[0]
        aa = torch.empty(n, 3, device=R.device)
        check(_lib.load().dyb_rotmat_to_aa_fwd(R.data_ptr(), aa.data_ptr(), n, stream_of(R)), "dyb_rotmat_to_aa_fwd")
        ctx.save_for_backward(R)
        return aa

    @staticmethod
    def backward(ctx, g):
        (R,) = ctx.saved_tensors
        n = R.shape[0]
        dR = torch.empty_like(R)
        g = g.contiguous().float()
        check(_lib.load().dyb_rotmat_to_aa_bwd(R.data_ptr(), g.data_ptr(), dR.data_ptr(), n, stream_of(R)),
              "dyb_rotmat_to_aa_bwd")
        return dR


def rotation_matrix_to_angle_axis(R):
    """(N,3,3) -> (N,3)."""
    return _R2AA.apply(R.reshape(-1, 3, 3))


def batch_rodrigues(theta):
    """(N,3) axis-angle -> (N,3,3), utils/geometry.py:9-24 (through the unit quaternion, 1e-8 added before the norm).  Its callers
    convert ground-truth poses (metric targets, retrieved exemplars): without a gradient to carry it is one launch of the library's
    rodrigues kernel (what the native stepper issues); an input that requires grad takes the torch composition below."""
    if not (torch.is_grad_enabled() and theta.requires_grad) and theta.dim() == 2 and theta.shape[1] == 3 and theta.numel() > 0:
        t = theta.detach().contiguous().float()
        R = torch.empty(t.shape[0], 3, 3, device=t.device)
        check(_lib.load().dyb_rodrigues_fwd(t.data_ptr(), R.data_ptr(), t.shape[0], stream_of(t)), "dyb_rodrigues_fwd")
        return R
    ang = (theta + 1e-8).norm(dim=1, keepdim=True)
    axis = theta / ang
    half = 0.5 * ang
    q = torch.cat([half.cos(), half.sin() * axis], 1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    rows = [w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
            2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
            2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]
    return torch.stack(rows, 1).view(-1, 3, 3)


class _Perspective(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, rotation, translation, focal, center):
        B, n = points.shape[0], points.shape[1]
        pts, rot, tr = points.contiguous().float(), rotation.contiguous().float(), translation.contiguous().float()
        cen = center.contiguous().float()
        out = torch.empty(B, n, 2, device=pts.device)
        ldf = 1 if focal.numel() == B and B > 1 else 0
        check(_lib.load().dyb_perspective_projection_fwd(pts.data_ptr(), rot.data_ptr(), tr.data_ptr(), focal.data_ptr(), ldf, cen.data_ptr(),
                                                         out.data_ptr(), B, n, stream_of(pts)), "dyb_perspective_projection_fwd")
        ctx.save_for_backward(pts, rot, tr, focal)
        ctx.ldf = ldf
        return out

    @staticmethod
    def backward(ctx, g2):
        pts, rot, tr, focal = ctx.saved_tensors
        B, n = pts.shape[0], pts.shape[1]
        g2 = g2.contiguous().float()
        dp, dt = torch.empty_like(pts), torch.empty_like(tr)
        check(_lib.load().dyb_perspective_projection_bwd(pts.data_ptr(), rot.data_ptr(), tr.data_ptr(), focal.data_ptr(), ctx.ldf,
                                                         g2.data_ptr(), dp.data_ptr(), dt.data_ptr(), B, n, stream_of(pts)),
              "dyb_perspective_projection_bwd")
        return dp, None, dt, None, None


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """utils/geometry.py:63-91, same signature: points (bs, N, 3), rotation (bs, 3, 3), translation (bs, 3), focal_length (bs,) or
    scalar, camera_center (bs, 2) -> (bs, N, 2).  Differentiable in points and translation (what base_adaptor.py:160-170
    differentiates); rotation, focal length and centre are treated as constants."""
    B = points.shape[0]
    f = torch.as_tensor(focal_length, dtype=torch.float32, device=points.device).reshape(-1)
    f = f.expand(B).contiguous() if f.numel() == 1 else f.contiguous()
    return _Perspective.apply(points, rotation, translation, f, camera_center)
