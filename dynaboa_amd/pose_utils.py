"""Procrustes alignment for PA-MPJPE (reference utils/pose_utils.py:9-64), batched NumPy.
Stays on the host like the reference's: 14x3 points per sample, one 3x3 SVD each."""
from __future__ import annotations

import numpy as np


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
