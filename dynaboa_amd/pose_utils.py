"""Procrustes alignment for PA-MPJPE (reference utils/pose_utils.py:9-64).

``pa_mpjpe_device`` is what the adaptation drivers use: the alignment and the error run in libdynaboa_hip.so
(csrc/losses.hip, one 3x3 SVD per sample) so the metric path ships scalars.  ``compute_similarity_transform_batch``
keeps the reference's NumPy surface (it is also the checker of the kernel's test)."""
from __future__ import annotations

import numpy as np
import torch


def pa_mpjpe_device(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """pred, gt: (n, J, 3) device tensors -> (n,) Procrustes-aligned mean per-joint error, same unit."""
    from . import _lib
    from ._abi import check
    from .hmr import stream_of
    pred, gt = pred.contiguous().float(), gt.contiguous().float()
    n, J = pred.shape[0], pred.shape[1]
    out = torch.empty(n, dtype=torch.float32, device=pred.device)
    check(_lib.load().dyb_pa_mpjpe(pred.data_ptr(), gt.data_ptr(), out.data_ptr(), None, n, J, stream_of(pred)), "dyb_pa_mpjpe")
    return out


def compute_similarity_transform_batch(S1: np.ndarray, S2: np.ndarray) -> np.ndarray:
    """Similarity-align every S1[i] (N,3) onto S2[i]; returns the aligned copies."""
    S1 = np.asarray(S1, np.float32)
    S2 = np.asarray(S2, np.float32)
    mu1, mu2 = S1.mean(1, keepdims=True), S2.mean(1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = (X1 ** 2).sum((1, 2))
    K = np.einsum("bni,bnj->bij", X1, X2)                    # 3x3 = X1^T X2 per sample
    U, s, Vh = np.linalg.svd(K)
    V = np.swapaxes(Vh, 1, 2)
    Z = np.tile(np.eye(3, dtype=S1.dtype), (S1.shape[0], 1, 1))
    Z[:, 2, 2] = np.sign(np.linalg.det(U @ Vh))
    R = V @ Z @ np.swapaxes(U, 1, 2)
    scale = np.einsum("bij,bji->b", R, K) / var1
    t = mu2 - scale[:, None, None] * (mu1 @ np.swapaxes(R, 1, 2))
    return scale[:, None, None] * (S1 @ np.swapaxes(R, 1, 2)) + t
