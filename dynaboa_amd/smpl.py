"""SMPL body model wrapper with the reference's surface (``model/smpl.py:15-37``):
``SMPL(model_dir, gender='neutral', create_transl=False, batch_size=1)``;
``forward(betas, body_pose, global_orient, pose2rot=True)`` -> object with ``.vertices (B,6890,3)``,
``.joints (B,49,3)`` (24 SMPL joints + 21 vertex joints + 9 regressed extras, gathered by the
49-entry joint map), ``.betas/.body_pose/.global_orient/.full_pose``.

The skinning itself (smplx ``lbs``) runs in libdynaboa_hip.so (csrc/smpl_lbs.hip) as one autograd
node with gradients to betas and the rotation matrices."""
from __future__ import annotations

import ctypes
import os
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, assets, constants as C
from ._abi import check
from .hmr import stream_of


def _host_ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


class _LBSFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, betas, rot, smpl):
        lib = _lib.load()
        B = betas.shape[0]
        betas = betas.float() if (betas.dim() == 2 and betas.stride(1) == 1) else betas.contiguous().float()
        rot = rot.contiguous().float()
        dev = betas.device
        verts = torch.empty(B, C.NUM_VERTS, 3, device=dev)
        joints = torch.empty(B, C.NUM_OUT_JOINTS, 3, device=dev)
        saved = torch.empty(int(lib.dyb_lbs_saved_floats(B)), device=dev)
        check(lib.dyb_lbs_fwd(smpl._pf, smpl._pi, betas.data_ptr(), betas.stride(0), rot.data_ptr(), verts.data_ptr(),
                              joints.data_ptr(), saved.data_ptr(), B, stream_of(betas)), "dyb_lbs_fwd")
        ctx.smpl = smpl
        ctx.set_materialize_grads(False)            # an unused output's gradient arrives as None (backward handles both)
        ctx.save_for_backward(rot, saved)
        return verts, joints

    @staticmethod
    def backward(ctx, d_verts, d_joints):
        lib = _lib.load()
        rot, saved = ctx.saved_tensors
        smpl = ctx.smpl
        B = rot.shape[0]
        dev = rot.device
        if d_joints is None:
            d_joints = torch.zeros(B, C.NUM_OUT_JOINTS, 3, device=dev)
        d_joints = d_joints.contiguous().float()
        dv_ptr = None
        if d_verts is not None:
            d_verts = d_verts.contiguous().float()
            dv_ptr = d_verts.data_ptr()
        drot = torch.empty(B, 24, 3, 3, device=dev)
        dbetas = torch.empty(B, 10, device=dev)
        wsb = int(lib.dyb_lbs_bwd_workspace_bytes(B))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        check(lib.dyb_lbs_bwd(smpl._pf, smpl._pi, rot.data_ptr(), saved.data_ptr(), d_joints.data_ptr(), dv_ptr,
                              drot.data_ptr(), dbetas.data_ptr(), 10, B, ws.data_ptr(), wsb, stream_of(rot)), "dyb_lbs_bwd")
        return dbetas, drot, None


def _rodrigues(rv: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues (angle = ||r + 1e-8||) in one launch.  Only the metric path converts
    the ground-truth axis-angle pose (reference dynaboa_benchmark.py:221-227); no gradient."""
    rv = rv.detach().contiguous().float()
    n = rv.shape[0]
    R = torch.empty(n, 3, 3, device=rv.device)
    check(_lib.load().dyb_rodrigues_fwd(rv.data_ptr(), R.data_ptr(), n, stream_of(rv)), "dyb_rodrigues_fwd")
    return R


class SMPL(nn.Module):
    def __init__(self, model_dir: Optional[str] = None, gender: str = "neutral", create_transl: bool = False,
                 batch_size: int = 1, tables: Optional[Dict[str, np.ndarray]] = None,
                 joint_regressor_extra: Optional[str] = None):
        super().__init__()
        if tables is None:
            if model_dir is None:
                raise ValueError("give either model_dir (SMPL_*.pkl as smplx expects) or tables=")
            pkl = os.path.join(model_dir, f"SMPL_{gender.upper()}.pkl")
            tables = assets.load_smpl_pkl(pkl, joint_regressor_extra or "data/J_regressor_extra.npy")
        if create_transl:
            raise NotImplementedError("the reference constructs SMPL with create_transl=False (base_adaptor.py:144-146)")
        Jr = tables["J_regressor"].astype(np.float64)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.register_buffer("v_template", f32(tables["v_template"]))
        self.register_buffer("shapedirs", f32(tables["shapedirs"].reshape(-1, C.NUM_BETAS)))
        self.register_buffer("posedirs", f32(tables["posedirs"]))
        self.register_buffer("weights_t", f32(tables["lbs_weights"].T))
        self.register_buffer("j_template", f32(Jr @ tables["v_template"].astype(np.float64)))
        self.register_buffer("j_shapedirs", f32(np.einsum("jv,vcl->jcl", Jr, tables["shapedirs"].astype(np.float64)).reshape(72, 10)))
        self.register_buffer("J_regressor_extra", f32(tables["J_regressor_extra"]))
        self.register_buffer("parents", torch.from_numpy(tables["parents"].astype(np.int32)))
        self.register_buffer("vertex_joint_ids", torch.tensor(C.VERTEX_JOINT_IDS, dtype=torch.int32))
        self.register_buffer("joint_map", torch.tensor(C.JOINT_MAP_49, dtype=torch.int32))
        self.faces = tables.get("faces")
        self._pf = self._pi = None
        self._refresh_pointers()

    def _refresh_pointers(self):
        self._keep_f, self._pf = _host_ptr_array([self.v_template, self.shapedirs, self.posedirs, self.weights_t,
                                                  self.j_template, self.j_shapedirs, self.J_regressor_extra])
        self._keep_i, self._pi = _host_ptr_array([self.parents, self.vertex_joint_ids, self.joint_map])

    def _apply(self, fn, *a, **k):          # .to(device) moves the tables: rebuild the pointer table
        out = super()._apply(fn, *a, **k)
        self._refresh_pointers()
        return out

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot: bool = True, **kwargs):
        B = betas.shape[0]
        if pose2rot:
            full = torch.cat([global_orient.reshape(B, -1, 3), body_pose.reshape(B, -1, 3)], 1)
            rot = _rodrigues(full.reshape(-1, 3)).view(B, 24, 3, 3)
        else:
            rot = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, 23, 3, 3)], 1)
            full = rot
        verts, joints = _LBSFunction.apply(betas, rot, self)
        return SimpleNamespace(vertices=verts, joints=joints, betas=betas, body_pose=body_pose,
                               global_orient=global_orient, full_pose=full)
