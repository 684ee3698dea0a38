"""Python handle of the native frame stepper (csrc/adapt_step.hip): one C call per adapted frame.

``Adaptor.adaptation`` (dynaboa_amd/benchmark.py, mirroring reference dynaboa_benchmark.py:126-193) forwards a frame
here when the configuration is one the stepper covers (first order, frame-loss set) - the same kernels in the same
order as the autograd composition, issued from C++.  torch still owns every buffer: theta, the Adam moments (shared with
``dynaboa_amd.optim.Adam``'s state so checkpoints / tests see them), the tables, the workspace blob and the record
buffers are tensors; the stepper only holds their addresses."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib
from ._abi import check
from .hmr import get_layout, stream_of


def supported(o) -> Optional[str]:
    """None if the stepper covers these options, else the reason it does not."""
    g = lambda k, d=0: getattr(o, k, d)
    if not g("use_boa", 1):
        return "use_boa=0"
    if g("second_order"):
        return "second order"
    if not (g("use_frame_losses_lower", 1) and g("use_frame_losses_upper", 1)):
        return "frame losses switched off"
    if g("use_temporal_losses_lower") or g("use_temporal_losses_upper"):
        return "temporal terms"
    if g("retrieval") or g("lower_level_mixtrain") or g("upper_level_mixtrain"):
        return "labelled exemplars"
    if g("dynamic_boa"):
        return "dynamic loop"
    if g("dump_predictions"):
        return "prediction dumps"
    if not g("share_forwards", 1) or not g("fused_level", 1):
        return "unshared / unfused schedule requested"
    return None


class NativeStepper:
    def __init__(self, adaptor, nframes: int):
        lib = self.lib = _lib.load()
        o = adaptor.options
        hmr = adaptor.model.module
        theta = hmr.theta
        dev = theta.device
        self.device = dev
        B = int(getattr(o, "batch_size", 1))
        self.B, self.K = B, int(o.inner_step)
        self.eval_lower = int(getattr(o, "eval_lower", 1))
        L = get_layout(B)
        h = ctypes.c_void_p()
        check(lib.dyb_stepper_create(L.plan, B, 224, 224, ctypes.byref(h)), "dyb_stepper_create")
        self.h = h
        self._keep = []
        si = lambda k, v: check(lib.dyb_stepper_set_i(h, k.encode(), int(v)), f"set_i {k}")
        sf = lambda k, v: check(lib.dyb_stepper_set_f(h, k.encode(), float(v)), f"set_f {k}")

        def sp(k, t):
            self._keep.append(t)
            check(lib.dyb_stepper_set_p(h, k.encode(), t.data_ptr()), f"set_p {k}")
        si("inner_step", self.K); si("eval_lower", self.eval_lower); si("n_iter", 3)
        si("use_side", 1 if getattr(adaptor, "_side", None) is not None else 0)
        for k in ("lr", "beta1", "beta2", "fastlr", "s2dloss_weight", "shape_prior_weight", "pose_prior_weight"):
            sf(k, getattr(o, k))
        opt = adaptor.optimizer
        sf("eps", opt.param_groups[0]["eps"])
        st = opt.state.get(theta)
        if st is None:
            st = opt.state[theta] = dict(step=0, exp_avg=torch.zeros_like(theta), exp_avg_sq=torch.zeros_like(theta))
        self._adam = st
        si("adam_step", st["step"])
        sp("theta", theta.data); sp("adam_m", st["exp_avg"]); sp("adam_v", st["exp_avg_sq"])
        sp("init_state", hmr.make_init_state(B))
        prior = adaptor.gmm_f
        sp("gmm_means", prior.means); sp("gmm_precisions", prior.precisions); sp("gmm_log_weights", prior.log_nll_weights)
        sp("j_regressor_h36m", adaptor.J_regressor.to(dev).contiguous())
        sp("j14", torch.tensor(adaptor.joint_mapper_h36m, dtype=torch.int32, device=dev))
        for name, smpl in (("neutral", adaptor.smpl_neutral), ("male", adaptor.smpl_male), ("female", adaptor.smpl_female)):
            for i, t in enumerate([smpl.v_template, smpl.shapedirs, smpl.posedirs, smpl.weights_t, smpl.j_template, smpl.j_shapedirs,
                                   smpl.J_regressor_extra]):
                sp(f"smpl_{name}_{i}", t)
            for i, t in enumerate([smpl.parents, smpl.vertex_joint_ids, smpl.joint_map]):
                sp(f"smpli_{name}_{i}", t)
        self.rec_floats = int(lib.dyb_stepper_get_i(h, b"record_floats"))
        self.loss_floats = int(lib.dyb_stepper_get_i(h, b"loss_floats"))
        self.slots_per_frame = (self.K if self.eval_lower else 0) + 1
        self.records = torch.zeros(max(1, nframes) * self.slots_per_frame, self.rec_floats, device=dev)
        self.loss_log = torch.zeros(max(1, nframes), self.loss_floats, device=dev)
        si("record_capacity", self.records.shape[0]); si("loss_capacity", self.loss_log.shape[0])
        sp("records", self.records); sp("loss_log", self.loss_log)
        nbytes = int(lib.dyb_stepper_workspace_bytes(h))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib.dyb_stepper_bind_workspace(h, self.ws.data_ptr(), nbytes, stream_of(theta)), "dyb_stepper_bind_workspace")
        self.frame = 0
        self._theta = theta
        # the weight-gradient stream is this stepper's own: several steppers (sequence replicas on one GPU) must not
        # serialise on one shared auxiliary stream
        self._aux = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dyb_stepper_destroy(self.h)
                self.h = None
        except Exception:      # noqa: BLE001
            pass

    def adapt_frame(self, batch: Dict[str, torch.Tensor], side_stream=None):
        """-> (frame index, first record slot).  Inputs must be contiguous fp32 (gender int64) device tensors."""
        f = self.frame
        if f >= self.loss_log.shape[0]:
            raise RuntimeError("native stepper: more frames than reset_records() announced")
        img, kp = batch["image"].contiguous().float(), batch["smpl_j2d"].contiguous().float()
        pose, betas = batch["pose"].contiguous().float(), batch["betas"].contiguous().float()
        gender = batch["gender"].contiguous().long()
        keep = (img, kp, pose, betas, gender)
        slot0 = f * self.slots_per_frame
        side = side_stream.cuda_stream if side_stream is not None else None
        if side_stream is not None:
            for t in keep:
                if t.is_cuda:
                    t.record_stream(side_stream)
        check(self.lib.dyb_stepper_adapt_frame(self.h, img.data_ptr(), kp.data_ptr(), pose.data_ptr(), betas.data_ptr(), gender.data_ptr(),
                                               slot0, f, stream_of(self._theta), self._aux.cuda_stream if self._aux is not None else None, side),
              "dyb_stepper_adapt_frame")
        self._adam["step"] = int(self.lib.dyb_stepper_get_i(self.h, b"adam_step"))
        self.frame += 1
        return f, slot0

    def join(self):
        check(self.lib.dyb_stepper_join(self.h, stream_of(self._theta)), "dyb_stepper_join")

    def record_views(self, slot: int):
        B = self.B
        r = self.records[slot]
        return dict(pred=r[:B * 42].view(B, 14, 3), gt=r[B * 42:B * 84].view(B, 14, 3), mpjpe=r[B * 84:B * 85], pve=r[B * 85])

    def losses(self, frame: int, level: int):
        """(s2d, shape prior, pose prior, weighted total) of level `level` (0..inner_step-1 lower, inner_step = upper)."""
        return self.loss_log[frame, 4 * level:4 * level + 4]
