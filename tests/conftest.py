import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.fixture(scope="session")
def gmm_t():
    from dynaboa_amd import assets
    return {k: torch.from_numpy(v) for k, v in assets.load_gmm_prior().items()}


@pytest.fixture(scope="session")
def smpl_tabs():
    from dynaboa_amd import assets
    return assets.make_synthetic_smpl(0)


@pytest.fixture(scope="session")
def ckpt_rand():
    """Seed-22 synthetic checkpoint with perturbed GN affine and a non-identity mean pose
    (the configuration goldens g3/g4/g5 were generated with)."""
    from dynaboa_amd import assets
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    return assets.make_synthetic_checkpoint(22, mp, randomize_norm=True, prefix="")["model"]
