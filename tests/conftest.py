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


# ---------------------------------------------------------------------------- fp32 noise floor of the g5 streams
def tensor_class(name):
    """(stage, kind) of a parameter: the granularity at which the noise floor is pooled (a single tensor's measured deviation is one
    draw of a random quantity; its class's largest deviation is a stable statistic)."""
    top = name.split(".")[0]
    # (the stem has ONE tensor per kind - no pool to speak of -: it shares layer1's classes, whose GroupNorm affines sit right behind it
    # and collect the same ReLU-flip noise of the whole network above)
    stage = "layer1" if top in ("conv1", "bn1") else top if top.startswith("layer") else "regressor"
    if top in ("conv1",) or ".conv" in name or "downsample.0" in name:
        kind = "conv"
    elif top == "bn1" or ".bn" in name or "downsample.1" in name:
        kind = "gn_" + name.rsplit(".", 1)[1]
    else:
        kind = "fc_" + name.rsplit(".", 1)[1] if "." in name else "fc"
    return stage, kind


NOISE_FACTOR = 3.0        # a bound = NOISE_FACTOR x the class's measured fp32-vs-fp64 floor
NOISE_MIN = {"nd": 5e-4, "cos": 2e-5}     # (never tighter than this: the golden's stored norms / slices are themselves one fp32 draw, and
#                                           the regressor's bias classes have five tensors - their measured floor is 5e-5 ... 1e-4)
# theta_after - theta_before (and the teacher's drift) are sums of ADAM-NORMALISED steps: an element whose gradient is rounding noise still
# moves by ~lr per step, with the sign of the noise.  One such element of a 256-element slice flipping its step moves the slice cosine by up
# to 2 / 256 whatever the implementation, so the slice bound of those two quantities is never tighter than that (a first-principles floor
# the three CPU draws - which share most of their kernels - underestimate; measured on the GPU's throughput schedule: 1.9e-3 on conv1)
ADAM_SLICE_FLIP = 2.0 / 256.0


def noise_bounds(tag, names, factor=NOISE_FACTOR):
    """Per-tensor bounds for the end-of-stream state of golden g5_<tag>, derived from tests/golden/g5_<tag>_noise.npz
    (tools/make_noise.py: the reference in fp32 and the oracle in fp32 - with and without oneDNN -, each against the oracle in fp64 on the
    same stream).
    -> {q: {"nd": array, "cos": array}} for q in m, v, d (and t = teacher drift): `nd` bounds the relative deviation of a tensor's
    norm from the golden's, `cos` is the LOWER bound of a slice cosine.  A tensor's bound is factor x the largest deviation either
    fp32 run shows over the tensor's class (stage x kind)."""
    z = golden(f"g5_{tag}_noise.npz")
    znames = [str(x) for x in z["names"]]
    idx = {n: i for i, n in enumerate(znames)}
    cls = [tensor_class(n) for n in znames]
    out = {}
    for q in ("m", "v", "d", "t"):
        if f"{q}_nd_ref" not in z.files:
            continue
        draws = [d for d in ("ref", "or", "o2") if f"{q}_nd_{d}" in z.files]      # the fp32 evaluations the file holds (tools/make_noise.py)
        nd = np.max([z[f"{q}_nd_{d}"] for d in draws], axis=0)
        cs = 1.0 - np.min([z[f"{q}_cos_{d}"] for d in draws], axis=0)
        cmax_nd, cmax_cs = {}, {}
        for i, c in enumerate(cls):
            cmax_nd[c] = max(cmax_nd.get(c, 0.0), float(nd[i]))
            cmax_cs[c] = max(cmax_cs.get(c, 0.0), float(cs[i]))
        cmin = ADAM_SLICE_FLIP if q in ("d", "t") else NOISE_MIN["cos"]
        out[q] = dict(nd=np.array([max(NOISE_MIN["nd"], factor * cmax_nd[tensor_class(n)]) for n in names]),
                      cos=np.array([1.0 - max(cmin, factor * cmax_cs[tensor_class(n)]) for n in names]),
                      floor_nd=np.array([nd[idx[n]] for n in names]))
    return out
