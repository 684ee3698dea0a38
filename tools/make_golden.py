#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported read-only from
/root/reference) on seeded inputs.  Runs only in the build container; the reference source
never leaves it - only inputs-by-seed and expected outputs are committed.

Third-party modules the reference imports but that are not installed (learn2learn, smplx, cv2,
torchvision, tensorboard, skimage, pyrender, trimesh, human_body_prior) are replaced by empty
stub modules so `import base_adaptor, dynaboa_benchmark` succeeds.  The two
whose *semantics* matter are supplied explicitly below:
  * MAML  : clone()/adapt() restated from learn2learn 0.1.5 on top of torch.func.functional_call
  * SMPL  : a callable returning .joints/.vertices computed by oracle.ref_cpu.smpl_forward on
            the seeded synthetic tables of dynaboa_amd.assets.make_synthetic_smpl
so the goldens pin every line of arithmetic that lives in /root/reference, driven end to end.

usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--only g1,g3] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from dynaboa_amd import assets, constants as C  # noqa: E402
from oracle import ref_cpu as O                 # noqa: E402


# ---------------------------------------------------------------------------- stubs
class _Stub(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def install_stubs():
    names = ["cv2", "learn2learn", "learn2learn.algorithms", "torchvision", "torchvision.transforms",
             "torchvision.models", "torchvision.models.resnet", "torch.utils.tensorboard", "skimage",
             "skimage.transform", "smplx", "smplx.utils", "smplx.lbs", "pyrender", "pyrender.constants",
             "pyrender.camera", "trimesh", "human_body_prior", "human_body_prior.tools",
             "human_body_prior.tools.model_loader"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _Stub(n)
    for n in names:                      # make `import a.b.c as x` resolve through attributes
        if "." in n:
            parent, child = n.rsplit(".", 1)
            if isinstance(sys.modules.get(parent), _Stub):
                setattr(sys.modules[parent], child, sys.modules[n])
    sys.modules["pyrender"].Camera = object
    sys.modules["pyrender.camera"].DEFAULT_Z_NEAR = 0.05
    sys.modules["learn2learn"].algorithms = sys.modules["learn2learn.algorithms"]
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_file(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ---------------------------------------------------------------------------- third-party semantics
class MAMLStub(torch.nn.Module):
    """learn2learn.algorithms.MAML semantics."""

    def __init__(self, module, lr, first_order=True, fast=None):
        super().__init__()
        self.module, self.lr, self.first_order = module, lr, first_order
        self._fast = fast

    def forward(self, *a, **k):
        if self._fast is None:
            return self.module(*a, **k)
        allp = dict(self._fast)
        allp.update(dict(self.module.named_buffers()))
        return torch.func.functional_call(self.module, allp, a, k)

    def clone(self):
        src = self._fast if self._fast is not None else dict(self.module.named_parameters())
        return MAMLStub(self.module, self.lr, self.first_order, {n: p.clone() for n, p in src.items()})

    def adapt(self, loss):
        so = not self.first_order
        names = list(self._fast)
        g = torch.autograd.grad(loss, [self._fast[n] for n in names], retain_graph=so, create_graph=so)
        self._fast = {n: self._fast[n] - self.lr * gi for n, gi in zip(names, g)}

    def parameters(self, recurse=True):
        return self.module.parameters() if self._fast is None else iter(self._fast.values())


class SMPLStub:
    def __init__(self, tables):
        self.T = tables

    def __call__(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, **kw):
        v, j = O.smpl_forward(self.T, betas, body_pose, global_orient, pose2rot=pose2rot)
        return types.SimpleNamespace(vertices=v, joints=j)


def t2n(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


SLICE_PARAMS = ["conv1.weight", "bn1.weight", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight",
                "layer2.0.conv2.weight", "layer2.3.bn3.bias", "layer3.0.downsample.0.weight",
                "layer3.5.conv1.weight", "layer4.0.conv2.weight", "layer4.2.conv3.weight",
                "layer4.2.bn3.weight", "fc1.weight", "fc1.bias", "fc2.weight", "decpose.weight",
                "decpose.bias", "decshape.weight", "deccam.bias"]


def head(t, n=256):
    return t.detach().flatten()[:n].double().numpy()


# ---------------------------------------------------------------------------- G1 geometry
def g1(out):
    geo = load_file("ref_geometry", "utils/geometry.py")
    g = torch.Generator().manual_seed(101)
    x6 = torch.randn(5, 144, generator=g)
    x6[0] = torch.tensor([1., 0, 0, 1, 0, 0]).repeat(24)            # identity
    x6 = x6.requires_grad_(True)
    R = geo.rot6d_to_rotmat(x6)
    wR = torch.randn(R.shape, generator=g)
    (gx6,) = torch.autograd.grad((R * wR).sum(), x6)

    aa = torch.randn(40, 3, generator=g) * 0.6
    aa[0] = 0.0
    aa[1] = torch.tensor([1e-4, -2e-4, 5e-5])
    aa[2] = torch.tensor([3.10, 0.2, -0.1])                          # near pi
    aa[3] = torch.tensor([0.0, 3.0, 0.5])
    aa[4] = torch.tensor([0.1, 0.2, 3.05])
    aa[5:12] *= 4.0                                                  # large angles -> other branches
    aa = aa.requires_grad_(True)
    Rr = geo.batch_rodrigues(aa)
    wRr = torch.randn(Rr.shape, generator=g)
    (gaa,) = torch.autograd.grad((Rr * wRr).sum(), aa)

    Rin = Rr.detach().clone().requires_grad_(True)
    back = geo.rotation_matrix_to_angle_axis(Rin)
    wb = torch.randn(back.shape, generator=g)
    (gRin,) = torch.autograd.grad((back * wb).sum(), Rin)

    np.savez_compressed(os.path.join(out, "g1_geometry.npz"),
                        x6=x6.detach().numpy(), rot6d_R=R.detach().numpy(), rot6d_w=wR.numpy(),
                        rot6d_gx=gx6.numpy(), aa=aa.detach().numpy(), rodrigues_R=Rr.detach().numpy(),
                        rodrigues_w=wRr.numpy(), rodrigues_gaa=gaa.numpy(),
                        r2aa_out=back.detach().numpy(), r2aa_w=wb.numpy(), r2aa_gR=gRin.numpy())
    print("g1 ok")


# ---------------------------------------------------------------------------- G2 GMM prior
def g2(out):
    prior = load_file("ref_prior", "utils/smplify/prior.py")
    P = prior.MaxMixturePrior(prior_folder=os.path.join(REF, "data"), num_gaussians=8, dtype=torch.float32)
    buf = assets.gmm_buffers_from_pickle(os.path.join(REF, "data", "gmm_08.pkl"))
    for k in ("means", "precisions", "nll_weights"):
        ref = getattr(P, k).numpy()
        assert np.allclose(ref, buf[k], rtol=1e-6, atol=0), k
    os.makedirs(os.path.join(ROOT, "dynaboa_amd", "assets"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "dynaboa_amd", "assets", "gmm_08_f32.npz"),
                        means=P.means.numpy(), precisions=P.precisions.numpy(),
                        nll_weights=P.nll_weights.numpy())
    g = torch.Generator().manual_seed(202)
    pose = (torch.randn(6, 69, generator=g) * 0.3).requires_grad_(True)
    betas = torch.zeros(6, 10)
    ll = P(pose, betas)
    (gp,) = torch.autograd.grad(ll.mean(), pose)
    np.savez_compressed(os.path.join(out, "g2_gmm.npz"), pose=pose.detach().numpy(),
                        ll=ll.detach().numpy(), grad_mean=gp.numpy())
    print("g2 ok")


# ---------------------------------------------------------------------------- G3 HMR forward/backward
def build_ref_hmr(ckpt_seed=22, randomize_norm=True, identity_pose=False):
    hm = load_file("ref_hmr", "model/hmr.py")
    mp = assets.make_smpl_mean_params(identity_pose=identity_pose, seed=3)
    tmp = tempfile.NamedTemporaryFile(suffix=".npz", delete=False)
    np.savez(tmp.name, **mp)
    model = hm.hmr(tmp.name)
    os.unlink(tmp.name)
    ck = assets.make_synthetic_checkpoint(ckpt_seed, mp, randomize_norm=randomize_norm, prefix="")
    model.load_state_dict(ck["model"], strict=True)
    return model.eval(), ck["model"]


def g3(out):
    model, sd = build_ref_hmr()
    fr = assets.make_frame(0, batch_size=2, seed=22)
    img = fr["image"]
    r, s, c, feats = model(img, need_feature=True)
    g = torch.Generator().manual_seed(303)
    wr, ws, wc = torch.randn(r.shape, generator=g), torch.randn(s.shape, generator=g), torch.randn(c.shape, generator=g)
    loss = (r * wr).sum() + (s * ws).sum() + (c * wc).sum()
    names = [n for n, _ in model.named_parameters()]
    grads = torch.autograd.grad(loss, list(model.parameters()))
    gd = dict(zip(names, grads))
    # n_iter / explicit init path (reference model/hmr.py:127,132-137)
    r1, s1, c1 = model(img[:1], init_pose=sd["init_pose"] * 0.9, init_shape=sd["init_shape"] + 0.1,
                       init_cam=sd["init_cam"] * 1.1, n_iter=2)
    np.savez_compressed(
        os.path.join(out, "g3_hmr.npz"),
        rotmat=r.detach().numpy(), shape=s.detach().numpy(), cam=c.detach().numpy(),
        feat5=feats[5].detach().numpy(), feat12=feats[12].detach().numpy(),
        feat_sum=np.array([float(f.double().sum()) for f in feats]),
        feat_abs=np.array([float(f.double().abs().sum()) for f in feats]),
        feat_shapes=np.array([list(f.shape) + [0] * (4 - f.dim()) for f in feats]),
        wr=wr.numpy(), ws=ws.numpy(), wc=wc.numpy(),
        grad_norms=np.array([float(gd[n].double().norm()) for n in names]),
        grad_names=np.array(names),
        **{"gs_" + n: head(gd[n]) for n in SLICE_PARAMS},
        alt_rotmat=r1.detach().numpy(), alt_shape=s1.detach().numpy(), alt_cam=c1.detach().numpy())
    print("g3 ok")


# ---------------------------------------------------------------------------- adaptor under test
def make_ref_adaptor(opts_over, identity_pose=False, randomize_norm=True, smpl_seed=0, first_order=True):
    import dynaboa_benchmark as DB          # reference module (stubs installed)
    prior = load_file("ref_prior", "utils/smplify/prior.py")
    opts = DB.parser.parse_args([])
    for k, v in opts_over.items():
        setattr(opts, k, v)
    opts.mixtrain = opts.lower_level_mixtrain or opts.upper_level_mixtrain
    a = DB.Adaptor.__new__(DB.Adaptor)
    a.options = opts
    a.device = torch.device("cpu")
    a.exppath = tempfile.mkdtemp()
    os.makedirs(os.path.join(a.exppath, "result"), exist_ok=True)
    model, sd = build_ref_hmr(randomize_norm=randomize_norm, identity_pose=identity_pose)
    a.model = MAMLStub(model, lr=opts.fastlr, first_order=first_order).eval()
    a.optimizer = torch.optim.Adam(a.model.parameters(), lr=opts.lr, betas=(opts.beta1, opts.beta2),
                                   foreach=False)
    if opts.use_meanteacher:
        teacher, _ = build_ref_hmr(randomize_norm=randomize_norm, identity_pose=identity_pose)
        for p in teacher.parameters():
            p.detach_()
        a.teacher = teacher
    a.gmm_f = prior.MaxMixturePrior(prior_folder=os.path.join(REF, "data"), num_gaussians=8, dtype=torch.float32)
    tabs = O.smpl_tables_to_torch(assets.make_synthetic_smpl(smpl_seed))
    a.smpl_neutral = SMPLStub(tabs)
    a.smpl_male = SMPLStub(O.smpl_tables_to_torch(assets.make_synthetic_smpl(smpl_seed + 1)))
    a.smpl_female = SMPLStub(O.smpl_tables_to_torch(assets.make_synthetic_smpl(smpl_seed + 2)))
    a.J_regressor = tabs["J_regressor_h36m"]
    a.joint_mapper_h36m = list(C.H36M_TO_J14)
    a.history, a.kp2dlosses_lower, a.kp2dlosses_upper, a.fit_losses = {}, [], {}, {}
    a.mpjpe_all_lower = [[] for _ in range(opts.inner_step)]
    a.pampjpe_all_lower = [[] for _ in range(opts.inner_step)]
    a.mpjpe_statistics, a.pampjpe_statistics = {}, {}
    a.feat_sims, a.optim_step_record = {}, []
    a.global_step = 0
    a.retrieval = lambda feature: assets.make_exemplars(a.global_step, opts.sample_num)
    return a, sd


# ---------------------------------------------------------------------------- G4 loss methods
def g4(out):
    a, sd = make_ref_adaptor(dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0))
    g = torch.Generator().manual_seed(404)
    B = 3
    aa = torch.randn(B * 24, 3, generator=g) * 0.3
    rot = O.smplx_rodrigues(aa).view(B, 24, 3, 3).requires_grad_(True)
    shape = (torch.randn(B, 10, generator=g) * 0.5).requires_grad_(True)
    cam = (torch.tensor([[0.9, 0.05, -0.03]]) + 0.05 * torch.randn(B, 3, generator=g)).requires_grad_(True)
    kp = assets.make_frame(7, B)["smpl_j2d"]
    so = a.decode_smpl_params(rot, shape)
    s2d = a.projection(cam, so["s3d"])["normed"]
    conf = kp[:, 25:, -1:].clone()
    l2d = (torch.nn.functional.mse_loss(s2d[:, 25:], kp[:, 25:, :-1], reduction="none") * conf).mean()
    lsh = a.cal_shape_prior(shape)
    lpo = a.cal_pose_prior(rot, shape)
    o = a.options
    loss = l2d * o.s2dloss_weight + lsh * o.shape_prior_weight + lpo * o.pose_prior_weight
    gr, gs, gc = torch.autograd.grad(loss, [rot, shape, cam], retain_graph=True)
    (gpo,) = torch.autograd.grad(lpo, rot, retain_graph=True)
    gj2d = torch.autograd.grad(l2d, [rot, shape, cam])
    gt3 = torch.randn(B, 24, 3, generator=g) * 0.3
    l3d = a.cal_s3d_loss(so["s3d"][:, 25:].detach(), gt3, conf)
    np.savez_compressed(os.path.join(out, "g4_losses.npz"), aa=aa.numpy(), shape=shape.detach().numpy(),
                        cam=cam.detach().numpy(), kp=kp.numpy(), s2d=s2d.detach().numpy(),
                        s3d=so["s3d"].detach().numpy(), l2d=float(l2d), lsh=float(lsh), lpo=float(lpo),
                        loss=float(loss), g_rot=gr.numpy(), g_shape=gs.numpy(), g_cam=gc.numpy(),
                        gpo_rot=gpo.numpy(), g2d_rot=gj2d[0].numpy(), g2d_shape=gj2d[1].numpy(),
                        g2d_cam=gj2d[2].numpy(), gt3=gt3.numpy(), l3d=float(l3d))
    print("g4 ok")


# ---------------------------------------------------------------------------- G5 adaptation stream
def record_exact_cosines(a):
    """Wrap the reference adaptor's cal_feature_diff (base_adaptor.py:211-219) so that every call ALSO leaves the 15 cosines evaluated in
    float64 on the same feature tensors (a.feat_sims64[step] = list of [15] arrays): F.cosine_similarity in fp32 on 2e5 ... 8e5
    elements carries a summation error of its own (a few 1e-6; it returns 1.000017 for feature 0), which a test must not mistake for a
    difference between implementations.  The reference's own values and decisions are untouched."""
    orig = a.cal_feature_diff
    a.feat_sims64 = {}

    def wrapped(fi, fj):
        out = orig(fi, fj)
        c = [float(torch.nn.functional.cosine_similarity(x.detach().double().flatten(), y.detach().double().flatten(), dim=0, eps=1e-12))
             for x, y in zip(fi, fj)]
        a.feat_sims64.setdefault(a.global_step, []).append(np.array(c))
        return out
    a.cal_feature_diff = wrapped


def gate_checks(a, step):
    """cosines of the 15 features at every check of the dynamic-BOA gate in frame `step` (the reference's self.feat_sims,
    dynaboa_benchmark.py:161-185): [checks][15] float64 holding the fp32 values .item() returned."""
    return np.array([[chk[i]["cos"] for i in range(len(chk))] for chk in a.feat_sims.get(step, [])], dtype=np.float64)


def run_stream(tag, out, opts_over, nframes, identity_pose=False, first_order=True, extra_payload=None):
    a, sd0 = make_ref_adaptor(opts_over, identity_pose=identity_pose, first_order=first_order)
    record_exact_cosines(a)
    names = [n for n, _ in a.model.module.named_parameters()]
    theta0 = {n: p.detach().clone() for n, p in a.model.module.named_parameters()}
    rec = dict(lower=[], upper=[], mpjpe=[], pampjpe=[], pve=[], steps=[])
    preds = []
    for step in range(nframes):
        a.global_step = step
        a.fit_losses = {}
        batch = assets.make_frame(step, 1, seed=22)
        a.model.eval()
        n_low0 = len(a.kp2dlosses_lower)
        mp, pa, pve = a.adaptation(batch)
        rec["mpjpe"].append(float(np.mean(mp))); rec["pampjpe"].append(float(np.mean(pa))); rec["pve"].append(float(pve))
        rec["lower"].append([float(x) for x in a.kp2dlosses_lower[n_low0:]])
        rec["upper"].append(float(a.fit_losses.get("ul/unlabelloss", float("nan"))))
        rec["steps"].append(a.optim_step_record[-1] if a.optim_step_record else 0)
        rec.setdefault("gate", []).append(gate_checks(a, step))
        with torch.no_grad():
            r, s, c = a.model(batch["image"])
            so = a.decode_smpl_params(r, s)
        preds.append(dict(rotmat=r.numpy(), shape=s.numpy(), cam=c.numpy(), joints=so["s3d"].numpy(),
                          vsum=np.array([float(so["vts"].double().sum()), float(so["vts"].double().abs().sum())])))
        if step == 0:
            # after the first Adam step exp_avg = (1 - beta1) * outer gradient: the gradient itself, per tensor
            st0 = a.optimizer.state
            pm0 = dict(zip(names, a.model.module.parameters()))
            first = dict(g1_norms=np.array([float(st0[pm0[n]]["exp_avg"].double().norm()) / (1 - opts_over.get("beta1", a.options.beta1))
                                            for n in names]))
            for n in SLICE_PARAMS:
                first["g1_" + n] = head(st0[pm0[n]]["exp_avg"]) / (1 - a.options.beta1)
    st = a.optimizer.state
    pmap = dict(zip(names, a.model.module.parameters()))
    payload = dict(
        nframes=nframes,
        lower2d=np.array([x + [np.nan] * (8 - len(x)) for x in rec["lower"]]),
        upper_loss=np.array(rec["upper"]), mpjpe=np.array(rec["mpjpe"]), pampjpe=np.array(rec["pampjpe"]),
        pve=np.array(rec["pve"]), extra_steps=np.array(rec["steps"]),
        delta_norms=np.array([float((pmap[n].detach().double() - theta0[n].double()).norm()) for n in names]),
        m_norms=np.array([float(st[pmap[n]]["exp_avg"].double().norm()) for n in names]),
        v_norms=np.array([float(st[pmap[n]]["exp_avg_sq"].double().norm()) for n in names]),
        adam_steps=int(st[pmap[names[0]]]["step"]), names=np.array(names))
    for n in SLICE_PARAMS:
        payload["d_" + n] = head(pmap[n].detach().double() - theta0[n].double())
        payload["m_" + n] = head(st[pmap[n]]["exp_avg"])
        payload["v_" + n] = head(st[pmap[n]]["exp_avg_sq"])
    payload.update(first)
    if a.options.dynamic_boa and a.options.use_boa:
        # every check of the gate: cos of the 15 features, NaN where the loop had already left; gate_1mcos12 = what the while
        # condition compares with the threshold (python double of 1 - fp32 cosine, dynaboa_benchmark.py:169)
        nchk = 1 + int(a.options.optim_steps)
        gc = np.full((nframes, nchk, 15), np.nan)
        for f, c in enumerate(rec["gate"]):
            gc[f, :len(c)] = c
        gc64 = np.full((nframes, nchk, 15), np.nan)
        for f in range(nframes):
            for k, c in enumerate(a.feat_sims64.get(f, [])):
                gc64[f, k] = c
        # gate_cos64: the same cosines in float64 on the same features (record_exact_cosines); gate_cos_fp32_noise: the largest
        # |fp32 - fp64| of feature 12 over the run = the reference's own summation noise in the quantity it thresholds
        payload.update(gate_cos=gc, gate_1mcos12=1.0 - gc[:, :, 12], gate_checks=np.array([len(c) for c in rec["gate"]]),
                       gate_threshold=np.array(float(a.options.cos_sim_threshold)), gate_cos64=gc64,
                       gate_cos_fp32_noise=np.array(float(np.nanmax(np.abs(gc[:, :, 12] - gc64[:, :, 12])))))
        print("gate: fp32-vs-fp64 cosine of feature 12, max |diff| %.2e (all features: %.2e)" %
              (float(payload["gate_cos_fp32_noise"]), float(np.nanmax(np.abs(gc - gc64)))))
    payload.update(extra_payload or {})
    for i, p in enumerate(preds):
        for k, v in p.items():
            payload[f"pred{i}_{k}"] = v
    if opts_over.get("use_meanteacher", 1):
        tmap = dict(a.teacher.named_parameters())
        payload["teacher_delta_norms"] = np.array(
            [float((tmap[n].detach().double() - theta0[n].double()).norm()) for n in names])
    np.savez_compressed(os.path.join(out, f"g5_{tag}.npz"), **payload)
    print(f"g5 {tag} ok", rec["upper"], rec["steps"])
    return payload


def g5(out):
    frame_only = dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0,
                      use_motion=0, dynamic_boa=0, use_temporal_losses_upper=0)
    run_stream("fo_inner3_frameonly", out, dict(frame_only, inner_step=3), 4)
    run_stream("fo_inner1_frameonly_identity", out, dict(frame_only, inner_step=1), 3, identity_pose=True)
    # the reference's full default term set (teacher + motion + labelled exemplars + dynamic loop)
    run_stream("fo_inner1_full", out, dict(inner_step=1, interval=2, optim_steps=2), 5)
    g5_forced(out)


def g5so(out):
    """Second-order MAML (learn2learn first_order=False: create_graph=True through every inner adapt) and its
    first-order twin on the same stream: the pair lets a test show an implementation follows the SO gradient and
    not merely the FO one (they differ by the alpha * Hessian-vector terms)."""
    frame_only = dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0,
                      use_motion=0, dynamic_boa=0, use_temporal_losses_upper=0)
    run_stream("so_inner2_frameonly", out, dict(frame_only, inner_step=2), 2, first_order=False)
    run_stream("fo_inner2_frameonly", out, dict(frame_only, inner_step=2), 2, first_order=True)


def g5so3(out):
    """Second order at the BENCHMARKED depth (inner_step=3, BASELINE configs[1]); its first-order twin is
    g5_fo_inner3_frameonly (same stream, same seeds)."""
    frame_only = dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0,
                      use_motion=0, dynamic_boa=0, use_temporal_losses_upper=0)
    run_stream("so_inner3_frameonly", out, dict(frame_only, inner_step=3), 2, first_order=False)


def g5so_full(out):
    """Second order (learn2learn first_order=False) with the reference's DEFAULT term set - labelled exemplars in the lower level,
    teacher + motion + exemplars in the upper (motion from frame 3 on at interval 2); its first-order twin is g5_fo_inner1_full
    (same stream, same seeds).  Pins --hvp exact --hvp_terms all (the multi-pass exact Hessian-vector products)."""
    run_stream("so_inner1_full", out, dict(inner_step=1, interval=2, optim_steps=2), 4, first_order=False)


def g5fo3(out):
    """The first-order twin of g5so3 alone (same call as in g5; regenerates g5_fo_inner3_frameonly incl. its g1_* keys)."""
    frame_only = dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0,
                      use_motion=0, dynamic_boa=0, use_temporal_losses_upper=0)
    run_stream("fo_inner3_frameonly", out, dict(frame_only, inner_step=3), 4)


def g5_forced(out):
    # lr=3e-6 moves features[12] by <1e-7 in cosine, so the dynamic loop of
    # dynaboa_benchmark.py:161-192 never fires above; a negative threshold forces the branch
    # (2 extra upper steps per frame, then the optim_steps cut-off).
    run_stream("fo_inner1_full_forced", out,
               dict(inner_step=1, interval=2, optim_steps=2, cos_sim_threshold=-1.0), 4)


def probe_gate(threshold, nframes, opts_over=None):
    """The reference's adaptation() at its LITERAL defaults (inner_step 1, interval 5, optim_steps 7, every term on) with the given
    gate threshold: per frame the list of 1 - cos(features[12]) the while condition saw, and the extra steps taken."""
    a, _ = make_ref_adaptor(dict(opts_over or {}, cos_sim_threshold=threshold))
    checks, steps = [], []
    for step in range(nframes):
        a.global_step = step
        a.fit_losses = {}
        a.model.eval()
        a.adaptation(assets.make_frame(step, 1, seed=22))
        checks.append(1.0 - gate_checks(a, step)[:, 12])
        steps.append(a.optim_step_record[-1])
    return checks, steps


def gate_margin(checks, threshold):
    """smallest relative distance of any check from the threshold: every decision of the run is safe against a perturbation of
    1 - cos smaller than margin * threshold."""
    return min(abs(float(d) - threshold) / threshold for c in checks for d in c)


G5_GATED_FRAMES = 10


def g5_gated(out, threshold=None, tag="fo_inner1_full_gated"):
    """the dynamic-BOA loop LEAVING BY CONVERGENCE (1 - cos <= threshold after >= 1 extra step,
    dynaboa_benchmark.py:161-192) at the literal defaults - interval 5 (motion term live from frame 5 on), optim_steps 7 - over 10
    frames.  The threshold is chosen ON THE REFERENCE RUN: candidates between the observed check values, each re-run gated (a
    decision changes everything after it), keeping the one whose run has the most distinct step counts with every decision at
    least 2 % of the threshold away from it (fp32 noise on 1 - cos ~ 1e-4 is ~1e-7, i.e. 0.1 %).  The chosen threshold, every
    check's 1 - cos and the margin are stored in the golden; `--gate_threshold` regenerates with a fixed value.

    Committed set (round 6; the search's candidates, thresholds as printed - the run is deterministic given the threshold):
      g5_fo_inner1_full_gated    --gate_threshold 1.910328865e-04   extra steps [6, 0, 0, 0, 0, 0, 0, 0, 0, 0]   margin 5.4 %
      g5_fo_inner1_full_gated_b  --gate_threshold 6.517768e-05      extra steps [8, 8, 8, 2, 0, 0, 0, 0, 8, 8]   margin 4.7 %
      g5_fo_inner1_full_gated_c  --gate_threshold 1.051724e-04      extra steps [8, 8, 8, 0, 0, 0, 0, 0, 8, 4]   margin 2.2 %
    (8 = the optim_steps cut-off: seven extra steps, then `break`; 1 ... 7 = left by convergence; 0 = never opened)"""
    n = G5_GATED_FRAMES
    log = []
    if threshold is None:
        forced, _ = probe_gate(-1.0, n)                     # loop forced open: 8 checks per frame
        vals = np.sort(np.concatenate(forced))
        print("forced-run 1-cos quantiles:", np.quantile(vals, [0, .1, .25, .5, .75, .9, 1]))
        best = None
        for q in (0.35, 0.5, 0.6, 0.7, 0.8, 0.25):
            t = float(np.quantile(vals, q))
            for _ in range(3):                               # nudge the candidate to the middle of the widest nearby gap of ITS run
                checks, steps = probe_gate(t, n)
                m = gate_margin(checks, t)
                log.append((t, m, steps))
                print(f"  candidate {t:.6e}: margin {m:.3%} steps {steps}")
                score = (len(set(steps)) >= 4 and max(steps) <= 7 and min(steps) == 0, m)
                if m >= 0.02 and (best is None or score > best[0]):
                    best = (score, t, checks, steps)
                if m >= 0.02:
                    break
                allv = np.sort(np.concatenate(checks))
                lo = allv[allv <= t].max() if (allv <= t).any() else t * 0.98
                hi = allv[allv > t].min() if (allv > t).any() else t * 1.02
                t = float(0.5 * (lo + hi))
            if best is not None and best[0][0] and best[0][1] >= 0.03:
                break
        assert best is not None, log
        threshold = best[1]
    checks, steps = probe_gate(threshold, n)
    margin = gate_margin(checks, threshold)
    print(f"g5 gated: threshold {threshold:.9e} margin {margin:.3%} extra steps {steps}")
    run_stream(tag, out, dict(inner_step=1, cos_sim_threshold=threshold), n,
               extra_payload=dict(gate_margin=np.array(margin),
                                  gate_search=np.array([[t, m] for t, m, _ in log]) if log else np.zeros((0, 2))))


# ---------------------------------------------------------------------------- G6 Procrustes
def g6(out):
    pu = load_file("ref_pose_utils", "utils/pose_utils.py")
    rng = np.random.default_rng(606)
    S1 = rng.normal(0, 0.3, (6, 14, 3)).astype(np.float32)
    S2 = rng.normal(0, 0.3, (6, 14, 3)).astype(np.float32)
    S2[1] = (S1[1] @ O.smplx_rodrigues(torch.tensor([[0.3, -0.8, 0.2]])).numpy()[0].T * 1.7 + 0.4).astype(np.float32)
    S2[2, :, 0] *= -1            # reflection case
    hat = pu.compute_similarity_transform_batch(S1, S2)
    np.savez_compressed(os.path.join(out, "g6_procrustes.npz"), S1=S1, S2=S2, S1_hat=hat)
    print("g6 ok")


# ---------------------------------------------------------------------------- G7 crop / keypoint preprocessing
def g7(out):
    """utils/dataprocess.py crop() + transform() and boa_dataset/pw3d.py j2d_processing arithmetic, run from the reference's
    own module.  skimage is not installed: its `resize` (imported by name into dataprocess) is replaced by the oracle's
    restatement of scikit-image 0.17.2's defaults, so the fixtures pin the box / paste / keypoint arithmetic to the reference
    and the resize to the restatement."""
    dp = load_file("ref_dataprocess", "utils/dataprocess.py")
    real = False
    try:                                        # the real scikit-image where it exists: then g7 pins the resize too
        import importlib
        for m in ("skimage", "skimage.transform"):
            sys.modules.pop(m, None)
        dp.resize = importlib.import_module("skimage.transform").resize
        real = True
    except Exception:      # noqa: BLE001
        dp.resize = O.skimage_resize
    print("g7 resize:", "scikit-image" if real else "oracle restatement (scikit-image absent: resize parity unpinned)")
    rng = np.random.default_rng(707)
    H, W = 150, 200
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 3 % 256)], -1).astype(np.uint8)
    img = (img.astype(np.int32) + rng.integers(-20, 20, img.shape)).clip(0, 255).astype(np.uint8)
    cases = [  # center (x, y), scale (box = 200 * scale px), res
        ((100.0, 75.0), 0.5, 64),       # 100 px box inside the frame: downscale 1.56
        ((100.0, 75.0), 0.9, 64),       # 180 px box, sticks out top / bottom: downscale 2.8
        ((20.0, 130.0), 0.6, 64),       # mostly outside (left / bottom)
        ((120.5, 60.25), 0.2, 64),      # 40 px box: upscale (no anti-aliasing)
        ((90.0, 70.0), 0.62, 224),      # the real output size, 124 px box: upscale 1.8
    ]
    payload = dict(img=img)
    for i, (c, s, res) in enumerate(cases):
        o = dp.crop(img.astype(np.float32).copy(), np.array(c), s, [res, res])
        ul = np.array(dp.transform([1, 1], np.array(c), s, [res, res], invert=1)) - 1
        br = np.array(dp.transform([res + 1, res + 1], np.array(c), s, [res, res], invert=1)) - 1
        payload.update({f"c{i}_center": np.array(c), f"c{i}_scale": np.array(s), f"c{i}_res": np.array(res),
                        f"c{i}_ul": ul, f"c{i}_br": br, f"c{i}_out": o.astype(np.float32)})
    payload["ncases"] = np.array(len(cases))
    kp = np.concatenate([rng.uniform(0, 200, (49, 2)), (rng.random((49, 1)) < 0.8).astype(float)], 1)
    kpo = kp.copy()
    for i in range(kpo.shape[0]):
        kpo[i, 0:2] = dp.transform(kpo[i, 0:2] + 1, np.array(cases[1][0]), cases[1][1], [224, 224], rot=0)
    kpo[:, :-1] = 2. * kpo[:, :-1] / 224 - 1.
    payload.update(kp=kp, kp_out=kpo.astype(np.float32))
    np.savez_compressed(os.path.join(out, "g7_preprocess.npz"), **payload)
    print("g7 ok", {k: v.shape for k, v in payload.items() if k.endswith("_out")})


# ---------------------------------------------------------------------------- G8 train-mode Dropout statistics
def g8(out):
    """The reference HMR left in train() mode (what its mean teacher is: base_adaptor.py:151-158 never calls .eval()): 512
    forwards of one frame with live nn.Dropout(0.5) after fc1 / fc2.  RNG streams cannot be matched across
    implementations, so the fixture is the DISTRIBUTION: per-output mean and standard deviation, plus the eval output."""
    model, sd = build_ref_hmr()
    img = assets.make_frame(0, 1, seed=22)["image"]
    with torch.no_grad():
        re, se, ce = model.eval()(img)
        model.train()
        torch.manual_seed(808)
        N = 512
        outs = [torch.cat([x.flatten() for x in model(img)]) for _ in range(N)]
    o = torch.stack(outs).double()
    np.savez_compressed(os.path.join(out, "g8_dropout.npz"), n=np.array(N), mean=o.mean(0).numpy(), std=o.std(0).numpy(),
                        eval=torch.cat([re.flatten(), se.flatten(), ce.flatten()]).numpy())
    print("g8 ok", float(o.std(0).mean()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g1,g2,g3,g4,g5,g6,g7,g8")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--gate_threshold", type=float, default=None, help="g5_gated: skip the search and use this threshold")
    ap.add_argument("--gate_tag", default="fo_inner1_full_gated", help="g5_gated: name of the golden (g5_<tag>.npz)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_stubs()
    for k in args.only.split(","):
        if k == "g5_gated":
            g5_gated(args.out, args.gate_threshold, args.gate_tag)
        else:
            globals()[k](args.out)
