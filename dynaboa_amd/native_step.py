"""Python handle of the native frame stepper (csrc/adapt_step.hip): one C call per adapted frame.

``Adaptor.adaptation`` (dynaboa_amd/benchmark.py, mirroring reference dynaboa_benchmark.py:126-193) forwards a frame
here when the configuration is one the stepper covers (first order, frame-loss set) - the same kernels in the same
order as the autograd composition, issued from C++.  torch still owns every buffer: theta, the Adam moments (shared with
``dynaboa_amd.optim.Adam``'s state so checkpoints / tests see them), the tables, the workspace blob and the record
buffers are tensors; the stepper only holds their addresses."""
from __future__ import annotations

import os

import ctypes
from typing import Dict, Optional

import torch

from . import _lib
from ._abi import check
from .hmr import get_layout, stream_of


def coverage(o, have_bundle: bool = True):
    """-> (mode, reason): mode 'frame' (first order, frame-loss set), 'full' (the reference's term set: teacher / motion /
    labelled exemplars / dynamic loop) or '' with `reason` saying why the stepper does not cover these options."""
    g = lambda k, d=0: getattr(o, k, d)
    if not g("use_boa", 1):
        return "", "use_boa=0"
    if g("second_order"):
        return "", "second order"
    if not (g("use_frame_losses_lower", 1) and g("use_frame_losses_upper", 1)):
        return "", "frame losses switched off"
    if g("dump_predictions"):
        return "", "prediction dumps"
    if not g("share_forwards", 1) or not g("fused_level", 1):
        return "", "unshared / unfused schedule requested"
    temporal = (g("use_temporal_losses_lower") or g("use_temporal_losses_upper")) and (g("use_meanteacher") or g("use_motion"))
    mix = g("lower_level_mixtrain") or g("upper_level_mixtrain")
    if not (temporal or mix or g("dynamic_boa") or g("use_meanteacher")):
        return "frame", None
    B = int(g("batch_size", 1))
    if B > 16:
        return "", "full term set natively needs batch <= 16"
    if mix and not g("retrieval"):
        return "", "labelled term without retrieval"
    if mix and int(g("sample_num", 1)) != B:
        return "", "sample_num != batch_size"
    if mix and not have_bundle and B != 1:
        return "", "feature-driven retrieval is batch 1"
    if mix and not have_bundle and bool(g("lower_level_mixtrain")) != bool(g("upper_level_mixtrain")):
        # the reference calls self.retrieval() - a draw from the seeded random.sample stream - at EVERY level when --retrieval=1
        # (base_adaptor.py:256,309); the stepper calls back only at levels with a labelled term, which consumes the stream at a
        # different rate when the two switches differ
        return "", "labelled term on one level only with feature-driven retrieval"
    return "full", None


def mode(o, have_bundle: bool = True) -> str:
    return coverage(o, have_bundle)[0]


def supported(o) -> Optional[str]:
    """None if the stepper covers these options, else the reason it does not."""
    return coverage(o)[1]


class NativeStepper:
    """One stepper for ONE adaptor, or - `adaptors` a list of S > 1 adaptors with identical options - for S independent
    sequence replicas stepped in lockstep: every launch of the chain covers all of them (csrc/dyb_common.h), each replica
    keeping its own weights / Adam state / records.  Their parameter arenas and Adam moments are moved into stacked
    [S][n] tensors (the adaptors' Parameters / optimizer state become views of the stacks)."""

    def __init__(self, adaptors, nframes: int):
        lib = self.lib = _lib.load()
        ads = list(adaptors) if isinstance(adaptors, (list, tuple)) else [adaptors]
        self.adaptors = ads
        S = self.S = len(ads)
        adaptor = ads[0]
        o = adaptor.options
        hmr = adaptor.model.module
        dev = hmr.theta.device
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
        si("replicas", S)
        self.mode = mode(o, getattr(adaptor, "bundle", None) is not None)
        self.full = self.mode == "full"
        si("inner_step", self.K); si("eval_lower", self.eval_lower); si("n_iter", 3)
        if self.full:
            g = lambda k, d=0: getattr(o, k, d)
            si("full", 1)
            si("temporal_lower", g("use_temporal_losses_lower")); si("temporal_upper", g("use_temporal_losses_upper", 1))
            si("use_teacher", g("use_meanteacher", 1)); si("use_motion", g("use_motion", 1)); si("interval", g("interval", 5))
            si("mix_lower", g("lower_level_mixtrain", 1)); si("mix_upper", g("upper_level_mixtrain", 1))
            si("dynamic", g("dynamic_boa", 1)); si("optim_steps", g("optim_steps", 7))
            for k in ("teacherloss_weight", "motionloss_weight", "labelloss_weight", "alpha", "cos_sim_threshold"):
                sf(k, getattr(o, k))
            self.dynamic, self.optim_steps = int(g("dynamic_boa", 1)), int(g("optim_steps", 7))
            # --teacher_dropout 1: the reference's teacher is never put in eval mode (base_adaptor.py:151-158), so its forwards run with live
            # nn.Dropout(0.5).  The stepper draws each teacher forward's masks from the same counter-based keys dynaboa_amd.hmr hands the
            # autograd path (seed = torch.initial_seed(), offset = the process-wide train-forward counter): the two paths are bit-identical
            self.teacher_train = bool(g("teacher_dropout")) and bool(g("use_meanteacher", 1))
            if self.teacher_train:
                si("teacher_train", 1)
                sf("drop_p", float(getattr(adaptor.teacher, "dropout_p", 0.5)))
        self.use_side = 1 if (S == 1 and getattr(adaptor, "_side", None) is not None) else 0
        si("use_side", self.use_side)
        # DYB_SIDE_THREAD=1: the side stream's launches (previous frame's final forward + record, ground-truth meshes) from a helper thread of
        # the library (csrc/adapt_step.hip "side_thread").  Off by default: measured on MI355X it changes nothing (93.7 frames/s either way,
        # host issue 10.41 ms per frame both: the one-sequence frame is bound by the device's chain of ~950 dependent kernels, the
        # calling thread merely keeps up with it - profiles/r04_sessions.txt, closing session)
        si("side_thread", 1 if (self.use_side and hmr.theta.is_cuda and os.environ.get("DYB_SIDE_THREAD", "0") == "1") else 0)
        for k in ("lr", "beta1", "beta2", "fastlr", "s2dloss_weight", "shape_prior_weight", "pose_prior_weight"):
            sf(k, getattr(o, k))
        sf("eps", adaptor.optimizer.param_groups[0]["eps"])
        # parameters and Adam moments: one [S][n] stack each; replica r's Parameter / optimizer state alias row r
        n = hmr.theta.numel()
        steps = []
        for a in ads:
            st = a.optimizer.state.get(a.model.module.theta)
            steps.append(0 if st is None else int(st["step"]))
        if S == 1:
            theta = hmr.theta
            st = adaptor.optimizer.state.get(theta)
            if st is None:
                st = adaptor.optimizer.state[theta] = dict(step=0, exp_avg=torch.zeros_like(theta), exp_avg_sq=torch.zeros_like(theta))
            self.theta, self.m, self.v = theta.data, st["exp_avg"], st["exp_avg_sq"]
            self._adam = [st]
        else:
            self.theta = torch.empty(S, n, device=dev)
            self.m, self.v = torch.zeros(S, n, device=dev), torch.zeros(S, n, device=dev)
            self._adam = []
            for r, a in enumerate(ads):
                p = a.model.module.theta
                self.theta[r].copy_(p.data)
                old = a.optimizer.state.get(p)
                if old is not None:
                    self.m[r].copy_(old["exp_avg"]); self.v[r].copy_(old["exp_avg_sq"])
                p.data = self.theta[r]
                a.optimizer.state[p] = dict(step=steps[r], exp_avg=self.m[r], exp_avg_sq=self.v[r])
                self._adam.append(a.optimizer.state[p])
        si("adam_step", steps[0])
        for r, t in enumerate(steps):                   # replicas may join with different histories (per-replica bias corrections)
            si(f"adam_step_{r}", t)
        sp("theta", self.theta); sp("adam_m", self.m); sp("adam_v", self.v)
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
        if self.full:
            self.slots_per_frame = int(lib.dyb_stepper_get_i(h, b"slots_per_frame"))
        cap = max(1, nframes)
        # records | loss log | gate log | pooled feature of every replica live in ONE [S][block] tensor: one replica arena for the
        # kernels' pointer rebasing instead of four (the arena table is kernel-argument state: csrc/dyb_common.h)
        n_rec, n_loss = cap * self.slots_per_frame * self.rec_floats, cap * self.loss_floats
        n_gate = cap * (1 + self.optim_steps) * 16 if self.full else 0
        n_feat = B * 2048 if self.full else 0
        pad64 = lambda n: (n + 63) // 64 * 64
        o_loss, o_gate, o_feat = pad64(n_rec), pad64(n_rec) + pad64(n_loss), pad64(n_rec) + pad64(n_loss) + pad64(n_gate)
        self.logs = torch.zeros(S, o_feat + pad64(n_feat), device=dev)
        self.records = self.logs[:, :n_rec].view(S, cap * self.slots_per_frame, self.rec_floats)
        self.loss_log = self.logs[:, o_loss:o_loss + n_loss].view(S, cap, self.loss_floats)
        sp("logs_base", self.logs)
        si("logs_bytes", self.logs.shape[1] * 4)
        si("record_capacity", cap * self.slots_per_frame); si("loss_capacity", cap)
        sp("records", self.records); sp("loss_log", self.loss_log)
        if self.full:
            if o.use_meanteacher:
                if S == 1:
                    sp("teacher", adaptor.teacher.theta.data)
                else:                                   # one [S][n] stack; replica r's teacher Parameter aliases row r
                    self.teacher = torch.empty(S, n, device=dev)
                    for r, a in enumerate(ads):
                        self.teacher[r].copy_(a.teacher.theta.data)
                        a.teacher.theta.data = self.teacher[r]
                    sp("teacher", self.teacher)
            self.gate_host = torch.zeros(16 * S).pin_memory() if dev.type == "cuda" else torch.zeros(16 * S)
            self.gate_log = self.logs[:, o_gate:o_gate + n_gate].view(S, cap, 1 + self.optim_steps, 16)
            self.feat5 = self.logs[:, o_feat:o_feat + n_feat].view(S, B, 2048)
            sp("gate_host", self.gate_host); sp("gate_log", self.gate_log); sp("feat5_out", self.feat5)
            self._cb = None
            if (o.lower_level_mixtrain or o.upper_level_mixtrain) and getattr(adaptor, "bundle", None) is None:
                # retrieval (base_adaptor.py:82-96) stays on the host: argmin over the cluster table + a seeded sample, per sequence
                self._ex_keep = [None] * S

                def _retrieve_one(r, out):
                    if dev.type == "cuda":
                        torch.cuda.current_stream(dev).synchronize()              # feat5 of this level's forward is complete
                    a = ads[r]
                    ex = a.retrieval(self.feat5[r])
                    a._last_h36m = ex
                    keep = [ex["img"].contiguous().float(), ex["keypoints"].contiguous().float(), ex["pose"].contiguous().float(),
                            ex["betas"].contiguous().float(), ex["pose_3d"].contiguous().float()]
                    self._ex_keep[r] = keep
                    for i, t in enumerate(keep):
                        out[i] = t.data_ptr()
                if S == 1:
                    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p))

                    def _retrieve(_user, _level, out):
                        try:
                            _retrieve_one(0, out)
                            return 0
                        except Exception as e:      # noqa: BLE001
                            self._cb_error = e
                            return 1
                    key = b"retrieve_fn"
                else:
                    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p))

                    def _retrieve(_user, _level, r, out):
                        try:
                            _retrieve_one(int(r), out)
                            return 0
                        except Exception as e:      # noqa: BLE001
                            self._cb_error = e
                            return 1
                    key = b"retrieve_rep_fn"
                self._cb = CB(_retrieve)
                check(lib.dyb_stepper_set_p(h, key, ctypes.cast(self._cb, ctypes.c_void_p)), "set_p " + key.decode())
        nbytes = int(lib.dyb_stepper_workspace_bytes(h))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib.dyb_stepper_bind_workspace(h, self.ws.data_ptr(), nbytes, stream_of(self.theta)), "dyb_stepper_bind_workspace")
        self.frame = 0
        # the weight-gradient stream is this stepper's own: several steppers (sequence replicas on one GPU) must not
        # serialise on one shared auxiliary stream
        self._aux = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        if os.environ.get("DYB_NO_AUX") == "1":      # diagnostic: every launch on the chain's stream (per-kernel durations without company)
            self._aux = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dyb_stepper_destroy(self.h)
                self.h = None
        except Exception:      # noqa: BLE001
            pass

    def output(self, which: int, replica: int = 0) -> torch.Tensor:
        """The last final inference of `replica` as a view of the stepper's workspace (dyb_stepper_output): which = 0 rotmat [B][24][9],
        1 state [B][160] (shape at 144, cam at 154), 2 vertices [B][6890][3], 3 joints [B][49][3]."""
        self.lib.dyb_stepper_output.restype = ctypes.c_void_p
        p = self.lib.dyb_stepper_output(self.h, int(which))
        if not p:
            raise RuntimeError("dyb_stepper_output: stepper not bound")
        shape = {0: (self.B, 24, 9), 1: (self.B, 160), 2: (self.B, 6890, 3), 3: (self.B, 49, 3)}[int(which)]
        off = int(p) - self.ws.data_ptr() + replica * (self.ws.numel() // self.S)
        n = 1
        for d in shape:
            n *= d
        if off < 0 or off % 4 or off + 4 * n > self.ws.numel():
            raise RuntimeError("dyb_stepper_output: pointer outside the workspace")
        return self.ws[off:off + 4 * n].view(torch.float32).view(shape)

    def adapt_frames(self, batches, side_stream=None):
        """One frame per replica (`batches`: list of S batch dicts).  -> (frame index, first record slot)."""
        f = self.frame
        if f >= self.loss_log.shape[1]:
            raise RuntimeError("native stepper: more frames than reset_records() announced")
        if len(batches) != self.S:
            raise ValueError(f"{self.S} replicas, {len(batches)} batches")
        cols = [[], [], [], [], []]
        for b in batches:
            for k, (key, conv) in enumerate((("image", torch.Tensor.float), ("smpl_j2d", torch.Tensor.float), ("pose", torch.Tensor.float),
                                             ("betas", torch.Tensor.float), ("gender", torch.Tensor.long))):
                cols[k].append(conv(b[key].contiguous()))
        keep = [t for c in cols for t in c]
        if side_stream is not None and self.use_side:
            for t in keep:
                if t.is_cuda:
                    t.record_stream(side_stream)
        slot0 = f * self.slots_per_frame
        side = side_stream.cuda_stream if (side_stream is not None and self.use_side) else None
        ptrs = (ctypes.c_void_p * (5 * self.S))(*[t.data_ptr() for t in keep])
        check(self.lib.dyb_stepper_adapt_frames(self.h, ctypes.cast(ptrs, ctypes.c_void_p), slot0, f, stream_of(self.theta),
                                                self._aux.cuda_stream if self._aux is not None else None, side),
              "dyb_stepper_adapt_frames")
        self._sync_adam_steps()
        # with a side stream the final inference of this frame is issued by the NEXT call (or join()): its inputs stay alive
        self._prev_keep, self._keep_inputs = getattr(self, "_keep_inputs", None), keep
        self.frame += 1
        return f, slot0

    def adapt_frame(self, batch: Dict[str, torch.Tensor], side_stream=None):
        return self.adapt_frames([batch], side_stream)

    def adapt_frame_full(self, batch, hist=None, exemplars=None):
        """One frame of the full term set.  hist = (image, kp2d) of the frame `interval` steps back or None; exemplars = dict
        (img, keypoints, pose, betas, pose_3d) or None when the retrieval callback supplies them.
        -> (frame index, first record slot, extra dynamic-loop steps)."""
        f = self.frame
        if f >= self.loss_log.shape[1]:
            raise RuntimeError("native stepper: more frames than reset_records() announced")
        c = lambda t: t if (t.dtype is torch.float32 and t.is_contiguous()) else t.contiguous().float()      # (no new view objects on the common path)
        keep = [c(batch["image"]), c(batch["smpl_j2d"]), c(batch["pose"]), c(batch["betas"]), batch["gender"].contiguous().long()]
        keep += [c(hist[0]), c(hist[1])] if hist is not None else [None, None]
        keep += [c(exemplars[k]) for k in ("img", "keypoints", "pose", "betas", "pose_3d")] if exemplars is not None else [None] * 5
        ptrs = (ctypes.c_void_p * 12)(*[None if t is None else t.data_ptr() for t in keep])
        extra = ctypes.c_int(0)
        slot0 = f * self.slots_per_frame
        self._cb_error = None
        self._drop_begin()
        rc = self.lib.dyb_stepper_adapt_frame_full(self.h, ctypes.cast(ptrs, ctypes.c_void_p), slot0, f, ctypes.cast(ctypes.pointer(extra), ctypes.c_void_p),
                                                   stream_of(self.theta), self._aux.cuda_stream if self._aux is not None else None)
        if getattr(self, "_cb_error", None) is not None:
            raise self._cb_error
        self._drop_end()
        check(rc, "dyb_stepper_adapt_frame_full")
        self._sync_adam_steps()
        self.frame += 1
        return f, slot0, int(extra.value)

    def _drop_begin(self):
        """train-mode teacher: hand the stepper this frame's dropout keys - torch's seed and the NEXT value of the process-wide
        train-forward counter (dynaboa_amd.hmr.next_dropout_key), exactly what the autograd path's teacher forwards would draw"""
        if getattr(self, "teacher_train", False):
            from . import hmr as _H
            seed = int(torch.initial_seed()) & ((1 << 64) - 1)
            check(self.lib.dyb_stepper_set_i(self.h, b"drop_seed", seed - (1 << 64) if seed >= (1 << 63) else seed), "set_i drop_seed")
            check(self.lib.dyb_stepper_set_i(self.h, b"drop_offset", _H._DROP_CALLS[0] + 1), "set_i drop_offset")

    def _drop_end(self):
        if getattr(self, "teacher_train", False):
            from . import hmr as _H
            _H._DROP_CALLS[0] += int(self.lib.dyb_stepper_get_i(self.h, b"drop_used"))      # the keys the frame's teacher forwards consumed

    def _sync_adam_steps(self):
        keys = getattr(self, "_adam_keys", None)
        if keys is None:
            keys = self._adam_keys = [f"adam_step_{r}".encode() for r in range(len(self._adam))]
        get = self.lib.dyb_stepper_get_i
        for st, k in zip(self._adam, keys):
            st["step"] = int(get(self.h, k))

    def set_active(self, idx=None):
        """The replicas the following frame steps cover (ascending indices; None = all): a sequence whose stream has ended leaves
        the set - its weights, Adam state and records stay as they are."""
        idx = [] if idx is None else sorted(int(i) for i in idx)
        arr = (ctypes.c_int * max(1, len(idx)))(*idx)
        check(self.lib.dyb_stepper_set_active(self.h, ctypes.cast(arr, ctypes.c_void_p), len(idx)), "dyb_stepper_set_active")
        self.active = list(range(self.S)) if not idx else idx

    def adapt_frames_full(self, batches, hists, exemplars):
        """One frame of the full term set per ACTIVE replica, in lockstep (lists of length S; entries of inactive replicas None).
        hists[r] = (image, kp2d) or None (all active replicas alike); exemplars[r] = dict or None (retrieval callback).
        -> (frame index, first record slot, [extra dynamic-loop steps per replica])."""
        f = self.frame
        if f >= self.loss_log.shape[1]:
            raise RuntimeError("native stepper: more frames than reset_records() announced")
        S = self.S
        c = lambda t: t if (t.dtype is torch.float32 and t.is_contiguous()) else t.contiguous().float()      # (no new view objects on the common path)
        ptrs = (ctypes.c_void_p * (12 * S))()
        keep = []
        for r in getattr(self, "active", range(S)):
            b = batches[r]
            row = [c(b["image"]), c(b["smpl_j2d"]), c(b["pose"]), c(b["betas"]), b["gender"].contiguous().long()]
            row += [c(hists[r][0]), c(hists[r][1])] if hists[r] is not None else [None, None]
            row += [c(exemplars[r][k]) for k in ("img", "keypoints", "pose", "betas", "pose_3d")] if exemplars[r] is not None else [None] * 5
            keep.append(row)
            for k, t in enumerate(row):
                ptrs[k * S + r] = None if t is None else t.data_ptr()
        extra = (ctypes.c_int * S)()
        slot0 = f * self.slots_per_frame
        self._cb_error = None
        self._drop_begin()
        rc = self.lib.dyb_stepper_adapt_frames_full(self.h, ctypes.cast(ptrs, ctypes.c_void_p), slot0, f, ctypes.cast(extra, ctypes.c_void_p),
                                                    stream_of(self.theta), self._aux.cuda_stream if self._aux is not None else None)
        if getattr(self, "_cb_error", None) is not None:
            raise self._cb_error
        self._drop_end()
        check(rc, "dyb_stepper_adapt_frames_full")
        self._sync_adam_steps()
        self._keep_inputs = keep
        self.frame += 1
        return f, slot0, [int(x) for x in extra]

    def level_row(self, frame: int, row: int, r: int = 0):
        """16-float log row `row` of `frame` (full mode): frame {s2d, shape, pose, total} | teacher {s2d, s3d, shape, pose, loss} |
        motion | labelled {s2d, s3d, shape, pose, loss} | level total."""
        return self.loss_log[r, frame, 16 * row:16 * row + 16]

    def gate_views(self, frame: int, r: int = 0):
        """(cos, means) of the dynamic-BOA gate's checks of `frame` for replica r: cos[k] = the 15 feature cosines of check k (views),
        means[k] = their sum / 14 (base_adaptor.py:218).  Cut / computed ONCE per frame for all replicas: per sequence this was 15 index
        operations and two tiny launches (a sum and a division) per check - at 32 sequences milliseconds of host time per step."""
        c = getattr(self, "_gate_cache", None)
        if c is None or c[0] != frame:
            g = self.gate_log[:, frame, :, :15]                                   # [S][checks][15]
            means = (g.sum(-1) / 14).unbind(0)                                    # S x [checks]
            cos = [[row.unbind(0) for row in rep.unbind(0)] for rep in g.unbind(0)]
            c = self._gate_cache = (frame, cos, [m.unbind(0) for m in means])
        return c[1][r], c[2][r]

    def join(self):
        check(self.lib.dyb_stepper_join(self.h, stream_of(self.theta)), "dyb_stepper_join")

    # The per-frame bookkeeping of S adaptors takes views of these buffers: ~45 indexing operations per sequence and frame when each
    # adaptor slices for itself - milliseconds of host time per step at 32 sequences, which the GPU spends idle at the frame boundary
    # (DESIGN.md 7).  The views of ALL sequences of a frame are therefore cut in a few batched operations on first use and handed
    # out from a one-frame cache; they are the same views of the same storage.
    def _frame_record_views(self, slot: int):
        T, B, S = self.slots_per_frame, self.B, self.S
        f = slot // T
        c = getattr(self, "_rec_cache", None)
        if c is None or c[0] != f:
            R = self.records[:, f * T:(f + 1) * T]                                # [S][T][record floats]
            cols = (R[..., :B * 42].view(S, T, B, 14, 3), R[..., B * 42:B * 84].view(S, T, B, 14, 3), R[..., B * 84:B * 85], R[..., B * 85])
            per = [[t.unbind(0) for t in col.unbind(0)] for col in cols]            # [field][replica][slot in frame]
            c = self._rec_cache = (f, per)
        return c[1]

    def record_views(self, slot: int, r: int = 0):
        per = self._frame_record_views(slot)
        k = slot % self.slots_per_frame
        return dict(pred=per[0][r][k], gt=per[1][r][k], mpjpe=per[2][r][k], pve=per[3][r][k])

    def losses(self, frame: int, level: int, r: int = 0):
        """(s2d, shape prior, pose prior, weighted total) of level `level` (0..inner_step-1 lower, inner_step = upper)."""
        if self.full:
            return self.loss_log[r, frame, 4 * level:4 * level + 4]
        c = getattr(self, "_loss_cache", None)
        if c is None or c[0] != frame:
            rows = self.loss_log[:, frame].view(self.S, -1, 4)                    # [S][levels][4]
            c = self._loss_cache = (frame, [t.unbind(0) for t in rows.unbind(0)])
        return c[1][r][level]


# Sequences per launch from which the throughput schedule (dy materialised once per layer by the one-pass GroupNorm backward, plain
# gradient convolutions on igemm_tp_kernel) beats the latency schedule: measured crossover, frames/s latency | throughput at
# 4: 173.6 | 169.3, 5: 185.8 | 198.9, 6: 198.4 | 222.9, 7: 206.1 | 247.9 (profiles/r05_sessions.txt s15; the library's own default
# is 8)
TP_MIN_SEQUENCES = int(os.environ.get("DYB_TP_MIN_SEQUENCES", "5"))       # (environment: crossover re-measurements, profiles/r06_sessions.txt s18)


def set_replica_policy(throughput: bool = True):
    """Process-wide launch policy for replica groups (library switches "rep_split" / "tp_min").  throughput = True: the split-K
    depth is chosen for the replica-multiplied grid and groups of >= TP_MIN_SEQUENCES run the throughput schedule - every sequence's
    results equal the sequence adapted alone to fp32 rounding (different summation orders; tests/test_adaptation_gpu.py
    test_replica_group_with_replica_aware_split).  False: the policy of a single sequence - bit-identical to sequences alone, at
    32 sequences roughly half the frame rate."""
    lib = _lib.load()
    check(lib.dyb_set_option(b"rep_split", 1 if throughput else 0), "dyb_set_option rep_split")
    if throughput:
        check(lib.dyb_set_option(b"tp_min", TP_MIN_SEQUENCES), "dyb_set_option tp_min")


class ReplicaGroup:
    """S independent sequences adapted on ONE GPU in lockstep (the shard axis of SURVEY 8e inside a device): S adaptors
    with identical options, one native stepper whose launches cover all of them.  ``step(batches)`` = one
    ``Adaptor.adaptation`` per replica, each on its own frame; records / losses land in each adaptor as if it had run
    alone (and its weights ARE what it would have computed alone: bit-identical, tests assert it)."""

    def __init__(self, adaptors, nframes: int):
        why = supported(adaptors[0].options)
        if why:
            raise ValueError(f"replica groups need a configuration the native stepper covers ({why})")
        for a in adaptors:
            a.reset_records(nframes)
            a._native_why = ""
        self.adaptors = list(adaptors)
        self.stepper = NativeStepper(self.adaptors, nframes)
        for r, a in enumerate(self.adaptors):
            a._native, a._native_replica = self.stepper, r

    def step(self, batches, global_step: int):
        """One frame per sequence.  batches[r] = None: sequence r has no frame left (streams of different lengths, reference
        boa_dataset/pw3d.py:19-35) - it sits this and all later steps out; the others are not held back by it."""
        ns = self.stepper
        active = [r for r, b in enumerate(batches) if b is not None]
        if not active:
            return [None] * len(batches)
        if len(active) != len(batches) or getattr(ns, "active", None) not in (None, active):
            ns.set_active(active)
        for r in active:
            a, b = self.adaptors[r], batches[r]
            a.global_step = global_step
            a.fit_losses = {}
            a.save_hist(b["image"], b["smpl_j2d"])
        out = [None] * len(batches)
        if ns.full:
            ins = {r: self.adaptors[r]._native_full_inputs(batches[r]) for r in active}
            hists = [ins[r][0] if r in ins else None for r in range(len(batches))]
            exs = [ins[r][1] if r in ins else None for r in range(len(batches))]
            f, slot, extra = ns.adapt_frames_full(batches, hists, exs)
            for r in active:
                out[r] = self.adaptors[r]._native_full_bookkeeping(f, slot, extra[r], hists[r])
        else:
            full = [b if b is not None else batches[active[0]] for b in batches]     # (inactive entries are ignored by the stepper)
            f, slot = ns.adapt_frames(full)
            for r in active:
                out[r] = self.adaptors[r]._native_bookkeeping(f, slot)
        return out

    def flush_metrics(self):
        from .benchmark import flush_metrics_of
        return flush_metrics_of(self.adaptors)
