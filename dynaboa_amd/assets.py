"""Synthetic and on-disk assets for the adaptation hot path.

The reference needs files that are not redistributable (``data/basemodel.pt``,
``data/smpl_mean_params.npz``, the SMPL ``.pkl`` models, ``J_regressor_extra.npy``,
``J_regressor_h36m.npy``; reference ``config.py:14-17``, ``base_adaptor.py:116,144-149``).
This module produces seeded stand-ins with exactly the on-disk schema and shapes so the
whole path can be exercised, timed and parity-checked without them, and
loaders for the real files when a user supplies them.

Everything here is host-side numpy / torch-CPU; nothing on the compute path.
"""
from __future__ import annotations

import math
import pickle
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import constants as C

RESNET50_BLOCKS = (3, 4, 6, 3)
RESNET50_PLANES = (64, 128, 256, 512)
NPOSE = 24 * 6
FC1_IN = 2048 + NPOSE + 13          # reference model/hmr.py:82


# --------------------------------------------------------------------------------------
# HMR parameter schema (reference model/hmr.py:63-125 registration order)
# --------------------------------------------------------------------------------------
def hmr_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of the 169 parameters in ``HMR.parameters()`` order, un-prefixed."""
    out: List[Tuple[str, Tuple[int, ...]]] = [("conv1.weight", (64, 3, 7, 7)),
                                             ("bn1.weight", (64,)), ("bn1.bias", (64,))]
    inplanes = 64
    for li, (nblk, planes) in enumerate(zip(RESNET50_BLOCKS, RESNET50_PLANES), start=1):
        for bi in range(nblk):
            p = f"layer{li}.{bi}."
            out += [(p + "conv1.weight", (planes, inplanes, 1, 1)),
                    (p + "bn1.weight", (planes,)), (p + "bn1.bias", (planes,)),
                    (p + "conv2.weight", (planes, planes, 3, 3)),
                    (p + "bn2.weight", (planes,)), (p + "bn2.bias", (planes,)),
                    (p + "conv3.weight", (planes * 4, planes, 1, 1)),
                    (p + "bn3.weight", (planes * 4,)), (p + "bn3.bias", (planes * 4,))]
            if bi == 0:
                out += [(p + "downsample.0.weight", (planes * 4, inplanes, 1, 1)),
                        (p + "downsample.1.weight", (planes * 4,)),
                        (p + "downsample.1.bias", (planes * 4,))]
            inplanes = planes * 4
    out += [("fc1.weight", (1024, FC1_IN)), ("fc1.bias", (1024,)),
            ("fc2.weight", (1024, 1024)), ("fc2.bias", (1024,)),
            ("decpose.weight", (NPOSE, 1024)), ("decpose.bias", (NPOSE,)),
            ("decshape.weight", (10, 1024)), ("decshape.bias", (10,)),
            ("deccam.weight", (3, 1024)), ("deccam.bias", (3,))]
    return out


def hmr_buffer_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    return [("init_pose", (1, NPOSE)), ("init_shape", (1, 10)), ("init_cam", (1, 3))]


def _rodrigues_np(aa: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = aa / np.maximum(th, 1e-12)
    K = np.zeros(aa.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def make_smpl_mean_params(identity_pose: bool = True, seed: int = 0) -> Dict[str, np.ndarray]:
    """Stand-in for ``smpl_mean_params.npz`` (keys ``pose,shape,cam``; reference
    ``model/hmr.py:100-103``).  ``identity_pose=True`` gives the 6-D identity
    ``[1,0,0,1,0,0]x24`` that BASELINE.md names; ``False`` draws a moderate random rest pose,
    closer to the real mean pose and better conditioned for the axis-angle prior."""
    if identity_pose:
        pose = np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 24)
    else:
        rng = np.random.default_rng(seed)
        R = _rodrigues_np(rng.normal(0, 0.25, (24, 3)))
        # reference reads the 6 numbers as a (3,2) matrix: columns 0,1 of R, row-major
        pose = R[:, :, :2].reshape(24 * 6).astype(np.float32)
    return {"pose": pose.astype(np.float32), "shape": np.zeros(10, np.float32),
            "cam": np.array([0.9, 0.0, 0.0], np.float32)}


def make_synthetic_checkpoint(seed: int = 22, mean_params: Dict[str, np.ndarray] | None = None,
                              randomize_norm: bool = False, prefix: str = "module."
                              ) -> Dict[str, Dict[str, torch.Tensor]]:
    """Seeded stand-in for ``data/basemodel.pt``: ``{'model': state_dict}`` whose keys carry
    the ``module.`` prefix of the MAML wrapper (reference ``base_adaptor.py:116-125``).

    Initialisation follows reference ``model/hmr.py:88-95``: convs N(0, sqrt(2/(k*k*Cout))),
    GroupNorm gamma=1 beta=0 (or perturbed when ``randomize_norm``, which parity tests use so a
    swapped gamma/beta index cannot hide), default ``nn.Linear`` init for fc1/fc2/biases,
    xavier-uniform gain 0.01 for the three decoder heads.
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in hmr_param_shapes():
        leaf = name.split(".")[-1]
        is_conv = len(shape) == 4
        is_norm = len(shape) == 1 and ("bn" in name or "downsample.1" in name)
        if is_conv:
            n = shape[2] * shape[3] * shape[0]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / n)
        elif is_norm:
            base = 1.0 if leaf == "weight" else 0.0
            t = torch.full(shape, base)
            if randomize_norm:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif name.startswith("dec") and leaf == "weight":
            fan_out, fan_in = shape
            a = 0.01 * math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        else:  # fc1/fc2 weights and every Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            fan_in = shape[1] if len(shape) == 2 else {"fc1": FC1_IN}.get(name.split(".")[0], 1024)
            a = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        sd[prefix + name] = t.float().contiguous()
    mp = mean_params or make_smpl_mean_params()
    sd[prefix + "init_pose"] = torch.from_numpy(mp["pose"]).float().reshape(1, NPOSE)
    sd[prefix + "init_shape"] = torch.from_numpy(mp["shape"]).float().reshape(1, 10)
    sd[prefix + "init_cam"] = torch.from_numpy(mp["cam"]).float().reshape(1, 3)
    return {"model": sd}


# --------------------------------------------------------------------------------------
# SMPL tables
# --------------------------------------------------------------------------------------
def make_synthetic_smpl(seed: int = 0, dense_skin: bool = False) -> Dict[str, np.ndarray]:
    """Seeded SMPL-shaped tables: true sizes, row-normalised regressors,
    <=4-sparse (or dense) skinning weights and the real kinematic tree.  float32."""
    rng = np.random.default_rng(1000 + seed)
    V, J = C.NUM_VERTS, C.NUM_SMPL_JOINTS
    v_template = (rng.normal(0, 1, (V, 3)) * np.array([0.25, 0.55, 0.12])).astype(np.float32)
    shapedirs = rng.normal(0, 0.012, (V, 3, C.NUM_BETAS)).astype(np.float32)
    posedirs = rng.normal(0, 0.004, (C.NUM_POSE_FEATS, V * 3)).astype(np.float32)

    def regressor(rows: int, nnz: int) -> np.ndarray:
        R = np.zeros((rows, V), np.float64)
        for r in range(rows):
            idx = rng.choice(V, nnz, replace=False)
            R[r, idx] = rng.random(nnz) + 0.05
        return (R / R.sum(1, keepdims=True)).astype(np.float32)

    J_regressor = regressor(J, 160)
    J_regressor_extra = regressor(C.NUM_EXTRA_JOINTS, 200)
    J_regressor_h36m = regressor(17, 200)
    if dense_skin:
        W = rng.random((V, J)) ** 6
    else:
        W = np.zeros((V, J))
        for v in range(V):
            idx = rng.choice(J, 4, replace=False)
            W[v, idx] = rng.random(4) + 0.02
    W = (W / W.sum(1, keepdims=True)).astype(np.float32)
    faces = rng.integers(0, V, (13776, 3)).astype(np.int64)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs,
                J_regressor=J_regressor, lbs_weights=W,
                parents=np.array(C.SMPL_PARENTS, np.int64),
                J_regressor_extra=J_regressor_extra, J_regressor_h36m=J_regressor_h36m,
                faces=faces)


def load_smpl_pkl(path: str, j_regressor_extra: str, j_regressor_h36m: str | None = None
                  ) -> Dict[str, np.ndarray]:
    """Read a real SMPL ``.pkl`` (the format ``smplx.SMPL`` consumes; reference
    ``model/smpl.py:18-21``, ``base_adaptor.py:144-149``) into the table dict used here."""
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    def dense(a):
        return np.asarray(a.todense() if hasattr(a, "todense") else a)
    sd = np.asarray(d["shapedirs"])[:, :, :C.NUM_BETAS]
    pd = np.asarray(d["posedirs"]).reshape(C.NUM_VERTS * 3, C.NUM_POSE_FEATS).T
    parents = np.asarray(d["kintree_table"])[0].astype(np.int64)
    parents[0] = -1
    out = dict(v_template=np.asarray(d["v_template"], np.float32),
               shapedirs=sd.astype(np.float32), posedirs=np.ascontiguousarray(pd, np.float32),
               J_regressor=dense(d["J_regressor"]).astype(np.float32),
               lbs_weights=np.asarray(d["weights"], np.float32), parents=parents,
               J_regressor_extra=np.load(j_regressor_extra).astype(np.float32),
               faces=np.asarray(d["f"]).astype(np.int64))
    if j_regressor_h36m:
        out["J_regressor_h36m"] = np.load(j_regressor_h36m).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------
# GMM pose prior
# --------------------------------------------------------------------------------------
def gmm_buffers_from_pickle(path: str) -> Dict[str, np.ndarray]:
    """Derive the three buffers ``merged_log_likelihood`` uses from ``gmm_08.pkl`` the way
    reference ``utils/smplify/prior.py:126-160`` does (precision = inv(float32 cov);
    nll_weights from the float64 covariance determinants normalised by their minimum)."""
    with open(path, "rb") as f:
        g = pickle.load(f, encoding="latin1")
    means = g["means"].astype(np.float32)
    covs32 = g["covars"].astype(np.float32)
    prec = np.stack([np.linalg.inv(c) for c in covs32]).astype(np.float32)
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in g["covars"]])
    const = (2 * np.pi) ** (69 / 2.0)
    nllw = np.asarray(g["weights"] / (const * (sqrdets / sqrdets.min()))).astype(np.float32)
    return dict(means=means, precisions=prec, nll_weights=nllw.reshape(1, -1))


def load_gmm_prior(path: str | None = None) -> Dict[str, np.ndarray]:
    """``path`` may be the original pickle or the float32 ``.npz`` re-export shipped in
    ``dynaboa_amd/assets/gmm_08_f32.npz`` (default)."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(__file__), "assets", "gmm_08_f32.npz")
    if path.endswith(".pkl"):
        return gmm_buffers_from_pickle(path)
    z = np.load(path)
    return {k: z[k] for k in ("means", "precisions", "nll_weights")}


# --------------------------------------------------------------------------------------
# Synthetic frames (SURVEY 8d "Synthetic inputs (concrete)")
# --------------------------------------------------------------------------------------
def make_frame(step: int, batch_size: int = 1, seed: int = 22) -> Dict[str, torch.Tensor]:
    """One test frame with the batch schema of reference ``boa_dataset/pw3d.py:115-124``."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + step)
    B = batch_size
    image = torch.randn(B, 3, C.IMG_RES, C.IMG_RES, generator=g)
    kp = torch.zeros(B, C.NUM_OUT_JOINTS, 3)
    kp[:, 25:, :2] = torch.rand(B, 24, 2, generator=g) * 2 - 1
    kp[:, 25:, 2] = (torch.rand(B, 24, generator=g) < 0.8).float()
    pose = torch.randn(B, 72, generator=g) * 0.2
    betas = torch.randn(B, 10, generator=g) * 0.5
    gender = (torch.rand(B, generator=g) < 0.5).long()
    return dict(image=image, smpl_j2d=kp, pose=pose, betas=betas, gender=gender)


_EX_DEVICE_CACHE: Dict[tuple, Dict[str, torch.Tensor]] = {}
EX_DEVICE_CACHE_STEPS = 2          # steps kept per (sample_num, device, seed): the current one and its predecessor


def make_exemplars_device(step: int, sample_num: int, device, seed: int = 22, keep_steps: int = EX_DEVICE_CACHE_STEPS) -> Dict[str, torch.Tensor]:
    """`make_exemplars` resident on `device`, shared per (step, sample_num, device): the synthetic exemplars depend on the step only, so
    the sequences of a replica group can share ONE generation and ONE upload per step instead of one per sequence.  OPT-IN
    (`synthetic_bundle(resident_exemplars=True)`, `bench.py --resident_exemplars 1`): it drops the per-sequence retrieval + upload
    work the reference does every level (base_adaptor.py:82-96), so records measured with it say "exemplars resident" (ADVICE r5).
    Bounded: only the last `keep_steps` steps of a (sample_num, device, seed) stay alive (4.8 MB each at sample_num 8).  The tensors are
    shared between callers, which must treat them as read-only (the stepper reads them through const pointers)."""
    fam = (int(sample_num), str(device), int(seed))
    key = (int(step),) + fam
    hit = _EX_DEVICE_CACHE.get(key)
    if hit is None:
        hit = _EX_DEVICE_CACHE[key] = {k: v.to(device) for k, v in make_exemplars(step, sample_num, seed).items()}
        mine = [k for k in _EX_DEVICE_CACHE if k[1:] == fam]
        for k in mine[:max(0, len(mine) - keep_steps)]:          # insertion order = age
            del _EX_DEVICE_CACHE[k]
    return dict(hit)


_EX_HOST_CACHE: Dict[tuple, Dict[str, torch.Tensor]] = {}
EX_HOST_CACHE_BYTES = 512 << 20     # pinned host memory the exemplar cache may hold (oldest entries leave first)


def make_exemplars_pinned(step: int, sample_num: int, seed: int = 22, max_bytes: int = EX_HOST_CACHE_BYTES) -> Dict[str, torch.Tensor]:
    """`make_exemplars` kept in (pinned, where CUDA is up) HOST memory: the stand-in for the reference's exemplar dataset living in host RAM
    (base_adaptor.py:55,82-96).  Every caller still pays its own host-to-device upload per retrieval (what the reference does per sequence
    and level); only the synthetic generation - which has no counterpart in the reference - is cached, and `pregenerate_exemplars` moves it
    out of a timed region altogether.  Pinning goes through torch's caching host allocator (an evicted entry's block is recycled, and the
    allocator itself keeps a block alive while an asynchronous copy from it is in flight).  Bounded by `max_bytes`."""
    key = (int(step), int(sample_num), int(seed))
    hit = _EX_HOST_CACHE.get(key)
    if hit is None:
        ex = make_exemplars(step, sample_num, seed)
        if torch.cuda.is_available():
            ex = {k: v.pin_memory() for k, v in ex.items()}
        hit = _EX_HOST_CACHE[key] = ex
        size = lambda e: sum(v.numel() * v.element_size() for v in e.values())
        total = sum(size(e) for e in _EX_HOST_CACHE.values())
        for k in list(_EX_HOST_CACHE):
            if total <= max_bytes or k == key:
                break
            total -= size(_EX_HOST_CACHE.pop(k))
    return dict(hit)


def pregenerate_exemplars(steps, sample_num: int, seed: int = 22) -> None:
    """Fill the pinned host cache for `steps` ahead of a timed loop: generation + pinning then cost nothing inside it, the per-retrieval
    upload still does."""
    for s in steps:
        make_exemplars_pinned(int(s), sample_num, seed)


def make_exemplars(step: int, sample_num: int = 1, seed: int = 22) -> Dict[str, torch.Tensor]:
    """Synthetic retrieved-exemplar batch with the keys of reference
    ``base_adaptor.py:347-351`` / ``SourceDataset.__getitem__`` (``:476-506``)."""
    g = torch.Generator().manual_seed(seed * 7_000_003 + step)
    S = sample_num
    kp = torch.zeros(S, C.NUM_OUT_JOINTS, 3)
    kp[:, 25:, :2] = torch.rand(S, 24, 2, generator=g) * 2 - 1
    kp[:, 25:, 2] = (torch.rand(S, 24, generator=g) < 0.8).float()
    p3 = torch.cat([torch.randn(S, 24, 3, generator=g) * 0.3, torch.ones(S, 24, 1)], -1)
    return dict(img=torch.randn(S, 3, C.IMG_RES, C.IMG_RES, generator=g),
                pose=torch.randn(S, 72, generator=g) * 0.2,
                betas=torch.randn(S, 10, generator=g) * 0.5, pose_3d=p3, keypoints=kp)
