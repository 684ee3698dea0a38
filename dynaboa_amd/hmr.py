"""HMR (ResNet-50/GroupNorm backbone + iterative SMPL-parameter regressor) on the native engine.

Python surface mirrors reference ``model/hmr.py``: ``hmr(smpl_mean_params, pretrained=False)`` builds
the module; ``HMR.forward(x, need_feature=False, init_pose=None, init_shape=None, init_cam=None,
n_iter=3)`` returns ``(pred_rotmat (B,24,3,3), pred_shape (B,10), pred_cam (B,3)[, features])``
(reference ``model/hmr.py:127-181``); ``state_dict()/load_state_dict()`` speak the reference's
parameter names and shapes.  All arithmetic happens in libdynaboa_hip.so: one C call per forward,
one per backward, hooked into torch.autograd as a single coarse node whose only differentiable
input is the flat parameter arena ``theta``.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, assets
from ._abi import check
from .hmr_layout import STATE_LD, HmrLayout

_LAYOUTS: Dict[tuple, HmrLayout] = {}
_WORKSPACES: Dict[tuple, torch.Tensor] = {}


def stream_of(t: torch.Tensor):
    """Raw hipStream_t of torch's current stream on t's device (None for host buffers, which only
    the emulator build used by the CPU tests accepts)."""
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


_AUX_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def aux_stream_of(t: torch.Tensor):
    """A per-device side stream for the engine's weight-gradient convolutions (None on host buffers)."""
    if not t.is_cuda:
        return None
    key = str(t.device)
    if key not in _AUX_STREAMS:
        _AUX_STREAMS[key] = torch.cuda.Stream(device=t.device)
    return _AUX_STREAMS[key].cuda_stream


def get_layout(batch: int, height: int = 224, width: int = 224) -> HmrLayout:
    key = (batch, height, width)
    if key not in _LAYOUTS:
        _LAYOUTS[key] = HmrLayout(_lib.load(), batch, height, width)
    return _LAYOUTS[key]


def get_workspace(L: HmrLayout, device: torch.device) -> torch.Tensor:
    """One engine workspace per (plan, device, stream): calls on different streams may overlap."""
    sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (L.B, L.H, L.W, str(device), sid)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = torch.empty(L.ws_bytes, dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


_BWD_STAGE: Dict[tuple, dict] = {}


def get_bwd_stage(L: HmrLayout, device: torch.device) -> dict:
    """Persistent per-(plan, device, stream) buffers for the backward call: the incoming gradients
    (tiny) are copied in and the parameter-gradient arena is written here and cloned out, so that the
    call's pointer arguments - the engine's graph-cache key - recur from frame to frame.  The arena's
    pad gaps are zeroed once; the engine overwrites every tensor span on each call."""
    sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (L.B, L.H, L.W, str(device), sid)
    st = _BWD_STAGE.get(key)
    if st is None:
        st = dict(d_state=torch.zeros(L.B, STATE_LD, dtype=torch.float32, device=device),
                  d_rot=torch.zeros(L.B, 24, 3, 3, dtype=torch.float32, device=device),
                  grads=torch.zeros(L.n_params, dtype=torch.float32, device=device))
        _BWD_STAGE[key] = st
    return st


def _feature_views(L: HmrLayout, acts: torch.Tensor, n_iter: int, train: bool = False):
    """The reference's feature list (model/hmr.py:139-168) as views of the activation arena.
    Spatial maps are exposed NCHW-shaped (channels-last strides) so shapes match the reference."""
    out = []
    feats = L.features_train if train else L.features
    for i in range(6 + 3 * n_iter):
        f = feats[i]
        d = [x for x in f["dims"] if x > 0]
        if len(d) == 4:
            v = acts[f["offset"]:f["offset"] + d[0] * d[1] * d[2] * d[3]].view(d[0], d[1], d[2], d[3]).permute(0, 3, 1, 2)
        else:
            v = torch.as_strided(acts, (d[0], d[1]), (f["row_stride"], 1), f["offset"])
        out.append(v)
    return out


_DROP_CALLS = [0]


def next_dropout_key():
    """(seed, offset) of the next train-mode forward: torch's global seed and a per-process call counter, so that
    torch.manual_seed() reproduces a run and every forward draws fresh masks."""
    _DROP_CALLS[0] += 1
    return int(torch.initial_seed()) & ((1 << 64) - 1), _DROP_CALLS[0]


class _HMRFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, image, init_state, n_iter, need_feature, drop=None):
        lib = _lib.load()
        B, _, H, W = image.shape
        L = get_layout(B, H, W)
        if theta.numel() != L.n_params:
            raise ValueError("parameter arena size does not match the engine plan")
        image = image.contiguous().float()
        init_state = init_state.contiguous().float()
        acts = torch.empty(L.act_floats, dtype=torch.float32, device=theta.device)
        ws = get_workspace(L, theta.device)
        if drop is None:
            check(lib.dyb_hmr_forward(L.plan, theta.data_ptr(), image.data_ptr(), init_state.data_ptr(), n_iter,
                                      acts.data_ptr(), ws.data_ptr(), L.ws_bytes, stream_of(theta)), "dyb_hmr_forward")
        else:       # train mode: nn.Dropout after fc1 / fc2 (reference model/hmr.py:165,169)
            check(lib.dyb_hmr_forward_train(L.plan, theta.data_ptr(), image.data_ptr(), init_state.data_ptr(), n_iter, acts.data_ptr(),
                                            ws.data_ptr(), L.ws_bytes, drop[0], drop[1], float(drop[2]), stream_of(theta)),
                  "dyb_hmr_forward_train")
        ctx.L, ctx.n_iter, ctx.drop = L, n_iter, drop
        ctx.save_for_backward(theta, acts)
        # outputs are views of the activation arena (kept alive by them; nobody writes it afterwards)
        rot = acts[L.off_rotmat:L.off_rotmat + B * 216].view(B, 24, 3, 3)
        st = acts[L.off_state:L.off_state + B * STATE_LD].view(B, STATE_LD)
        shape, cam = st[:, 144:154], st[:, 154:157]
        feats = tuple(_feature_views(L, acts, n_iter, train=drop is not None)) if need_feature else ()
        ctx.mark_non_differentiable(*feats)
        return (rot, shape, cam) + feats

    @staticmethod
    def backward(ctx, d_rot, d_shape, d_cam, *_unused):
        lib = _lib.load()
        theta, acts = ctx.saved_tensors
        L = ctx.L
        B = L.B
        stage = get_bwd_stage(L, theta.device)
        d_state, d_rot_s, grads = stage["d_state"], stage["d_rot"], stage["grads"]
        if d_shape is not None:
            d_state[:, 144:154].copy_(d_shape)
        else:
            d_state[:, 144:154].zero_()
        if d_cam is not None:
            d_state[:, 154:157].copy_(d_cam)
        else:
            d_state[:, 154:157].zero_()
        if d_rot is not None:
            d_rot_s.copy_(d_rot)
        else:
            d_rot_s.zero_()
        ws = get_workspace(L, theta.device)
        if ctx.drop is None:
            check(lib.dyb_hmr_backward(L.plan, theta.data_ptr(), acts.data_ptr(), d_rot_s.data_ptr(), d_state.data_ptr(),
                                       ctx.n_iter, grads.data_ptr(), ws.data_ptr(), L.ws_bytes, stream_of(theta),
                                       aux_stream_of(theta)),
                  "dyb_hmr_backward")
        else:
            d = ctx.drop
            check(lib.dyb_hmr_backward_train(L.plan, theta.data_ptr(), acts.data_ptr(), d_rot_s.data_ptr(), d_state.data_ptr(),
                                             ctx.n_iter, grads.data_ptr(), ws.data_ptr(), L.ws_bytes, d[0], d[1], float(d[2]),
                                             stream_of(theta), aux_stream_of(theta)), "dyb_hmr_backward_train")
        return grads.clone(), None, None, None, None, None


def hmr_apply(theta, image, init_state, n_iter=3, need_feature=False, drop=None):
    """Functional form used by the MAML learner: forward with an explicit parameter arena.  drop = (seed, offset, p):
    train mode (Dropout after fc1 / fc2)."""
    out = _HMRFunction.apply(theta, image, init_state, n_iter, need_feature, drop)
    if need_feature:
        return out[0], out[1], out[2], list(out[3:])
    return out[0], out[1], out[2]


_INIT_CACHE: dict = {}      # (seed, mean-parameter hashes) -> (packed arena, init_pose, init_shape, init_cam), one entry
_PACK_CACHE: dict = {}      # the last complete checkpoint packed: keep (its tensors), sig (addresses + versions), packed


class HMR(nn.Module):
    """SMPL iterative regressor with ResNet-50(GroupNorm) backbone; parameters live in one arena."""
    dropout_p = 0.5        # nn.Dropout() default, reference model/hmr.py:84,86

    def __init__(self, smpl_mean_params, seed: Optional[int] = None):
        super().__init__()
        if isinstance(smpl_mean_params, (str, bytes)):
            mp = np.load(smpl_mean_params)
            mp = {k: mp[k] for k in ("pose", "shape", "cam")}
        else:
            mp = smpl_mean_params
        self._layout1 = get_layout(1)
        mpf = {k: np.asarray(v, np.float32) for k, v in mp.items()}
        # a seeded initialisation is a pure function of (seed, mean parameters): drivers that build many models (sequence
        # replicas, the bench's side runs) get it from a one-entry cache instead of 0.15 s of torch.randn + packing each
        key = None if seed is None else (int(seed),) + tuple(hash(mpf[k].tobytes()) for k in ("pose", "shape", "cam"))
        hit = _INIT_CACHE.get(key) if key is not None else None
        if hit is None:
            sd = assets.make_synthetic_checkpoint(seed if seed is not None else int(torch.randint(0, 2**31 - 1, (1,))), mpf, prefix="")["model"]
            hit = (self._layout1.pack(sd), sd["init_pose"].clone(), sd["init_shape"].clone(), sd["init_cam"].clone())
            if key is not None:
                _INIT_CACHE.clear()
                _INIT_CACHE[key] = hit
        self.theta = nn.Parameter(hit[0].clone())
        self.register_buffer("init_pose", hit[1].clone())
        self.register_buffer("init_shape", hit[2].clone())
        self.register_buffer("init_cam", hit[3].clone())

    # ---- reference-named checkpoint I/O --------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False, **kw):
        out = OrderedDict() if destination is None else destination
        ref = self._layout1.unpack(self.theta.data)
        for name, _ in assets.hmr_param_shapes():
            out[prefix + name] = ref[name].to(self.theta.device)
        for b in ("init_pose", "init_shape", "init_cam"):
            out[prefix + b] = getattr(self, b).detach().clone()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        want = [n for n, _ in assets.hmr_param_shapes()] + ["init_pose", "init_shape", "init_cam"]
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for HMR: missing {missing[:5]} unexpected {unexpected[:5]}")
        # the same checkpoint dict loaded again (every sequence replica / side run loads the bundle's): its packed arena is kept,
        # keyed on the tensors' storage addresses and versions (the tensors are kept alive by the cache entry)
        sig = tuple((k, state_dict[k].data_ptr(), state_dict[k]._version) for k in want) if not missing else None
        if sig is not None and _PACK_CACHE.get("sig") == sig:          # (the cache keeps those tensors alive: their addresses cannot be reused)
            packed, sd = _PACK_CACHE["packed"], {b: state_dict[b].detach().cpu() for b in ("init_pose", "init_shape", "init_cam")}
        else:
            sd = {k: v.detach().cpu() for k, v in state_dict.items()}
            if missing:
                cur = self.state_dict()
                for k in missing:
                    sd[k] = cur[k].cpu()
            packed = self._layout1.pack(sd)
            if sig is not None:
                _PACK_CACHE.update(keep=[state_dict[k] for k in want], sig=sig, packed=packed)
        with torch.no_grad():
            self.theta.copy_(packed.to(self.theta.device))
            for b in ("init_pose", "init_shape", "init_cam"):
                getattr(self, b).copy_(sd[b].reshape(getattr(self, b).shape).to(self.theta.device))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def named_reference_parameters(self):
        """The 169 reference-shaped parameter tensors (copies) in ``HMR.parameters()`` order."""
        ref = self._layout1.unpack(self.theta.data)
        return [(n, ref[n]) for n, _ in assets.hmr_param_shapes()]

    # ---- forward ---------------------------------------------------------------------------------
    def make_init_state(self, batch_size, init_pose=None, init_shape=None, init_cam=None):
        dev = self.theta.device
        if init_pose is None and init_shape is None and init_cam is None:
            key = (batch_size, str(dev), self.init_pose._version, self.init_shape._version, self.init_cam._version)
            cache = self.__dict__.setdefault("_init_cache", {})
            if key not in cache:
                cache.clear()
                cache[key] = self.make_init_state(batch_size, self.init_pose.expand(batch_size, -1))
            return cache[key]
        st = torch.zeros(batch_size, STATE_LD, dtype=torch.float32, device=dev)
        st[:, :144] = self.init_pose.expand(batch_size, -1) if init_pose is None else init_pose
        st[:, 144:154] = self.init_shape.expand(batch_size, -1) if init_shape is None else init_shape
        st[:, 154:157] = self.init_cam.expand(batch_size, -1) if init_cam is None else init_cam
        return st

    def forward(self, x, need_feature=False, init_pose=None, init_shape=None, init_cam=None, n_iter=3, theta=None):
        st = self.make_init_state(x.shape[0], init_pose, init_shape, init_cam)
        # train(): the two nn.Dropout() layers of reference model/hmr.py:84,86 are live (p = 0.5, fresh masks per call)
        drop = next_dropout_key() + (self.dropout_p,) if self.training else None
        return hmr_apply(self.theta if theta is None else theta, x, st, n_iter, need_feature, drop)


def hmr(smpl_mean_params, pretrained: bool = False, **kwargs) -> HMR:
    """Constructor with the reference's signature (model/hmr.py:314-323)."""
    if pretrained:
        raise NotImplementedError("ImageNet initialisation needs torchvision weights; load a checkpoint instead")
    return HMR(smpl_mean_params, **kwargs)
