"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the reference's
own code (tools/make_golden.py).  Runs on CPU; no HIP involved."""
import numpy as np
import pytest
import torch

from conftest import cosine, golden, rel_err
from oracle import ref_cpu as O

SLICE_PARAMS = ["conv1.weight", "bn1.weight", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight",
                "layer2.0.conv2.weight", "layer2.3.bn3.bias", "layer3.0.downsample.0.weight",
                "layer3.5.conv1.weight", "layer4.0.conv2.weight", "layer4.2.conv3.weight",
                "layer4.2.bn3.weight", "fc1.weight", "fc1.bias", "fc2.weight", "decpose.weight",
                "decpose.bias", "decshape.weight", "deccam.bias"]


def test_g1_rot6d():
    g = golden("g1_geometry.npz")
    x = torch.from_numpy(g["x6"]).requires_grad_(True)
    R = O.rot6d_to_rotmat(x)
    assert rel_err(R.detach(), g["rot6d_R"]) < 1e-6
    (gx,) = torch.autograd.grad((R * torch.from_numpy(g["rot6d_w"])).sum(), x)
    assert rel_err(gx, g["rot6d_gx"]) < 1e-5


def test_g1_rodrigues_and_inverse():
    g = golden("g1_geometry.npz")
    aa = torch.from_numpy(g["aa"]).requires_grad_(True)
    R = O.batch_rodrigues(aa)
    assert rel_err(R.detach(), g["rodrigues_R"]) < 1e-6
    (ga,) = torch.autograd.grad((R * torch.from_numpy(g["rodrigues_w"])).sum(), aa)
    assert rel_err(ga, g["rodrigues_gaa"]) < 1e-5
    Rin = torch.from_numpy(g["rodrigues_R"]).requires_grad_(True)
    back = O.rotmat_to_axis_angle(Rin)
    np.testing.assert_allclose(back.detach().numpy(), g["r2aa_out"], rtol=1e-5, atol=1e-6)
    (gR,) = torch.autograd.grad((back * torch.from_numpy(g["r2aa_w"])).sum(), Rin)
    # rows 0 (theta=0) and 1 (theta~1e-4) are singular in the reference's own autograd
    ref = g["r2aa_gR"]
    ok = np.isfinite(ref).all(axis=(1, 2))
    ok[:2] = False
    assert rel_err(gR.numpy()[ok], ref[ok]) < 1e-4


def test_g2_gmm(gmm_t):
    g = golden("g2_gmm.npz")
    pose = torch.from_numpy(g["pose"]).requires_grad_(True)
    ll = O.gmm_prior(pose, gmm_t)
    assert rel_err(ll.detach(), g["ll"]) < 1e-6
    (gp,) = torch.autograd.grad(ll.mean(), pose)
    assert rel_err(gp, g["grad_mean"]) < 1e-5


@pytest.mark.slow
def test_g3_hmr_forward_backward(ckpt_rand):
    from dynaboa_amd import assets
    g = golden("g3_hmr.npz")
    P = {k: v.clone().requires_grad_(not k.startswith("init_")) for k, v in ckpt_rand.items()}
    img = assets.make_frame(0, batch_size=2, seed=22)["image"]
    r, s, c, feats = O.hmr_forward(P, img, need_feature=True)
    assert rel_err(r.detach(), g["rotmat"]) < 1e-5
    assert rel_err(s.detach(), g["shape"]) < 1e-5
    assert rel_err(c.detach(), g["cam"]) < 1e-5
    assert rel_err(feats[5].detach(), g["feat5"]) < 1e-5
    assert rel_err(feats[12].detach(), g["feat12"]) < 1e-5
    assert len(feats) == 15
    for i, f in enumerate(feats):
        assert list(f.shape) == [d for d in g["feat_shapes"][i] if d > 0]
        assert abs(float(f.double().abs().sum()) - g["feat_abs"][i]) < 1e-5 * g["feat_abs"][i]
    loss = (r * torch.from_numpy(g["wr"])).sum() + (s * torch.from_numpy(g["ws"])).sum() \
        + (c * torch.from_numpy(g["wc"])).sum()
    names = [str(n) for n in g["grad_names"]]
    grads = dict(zip(names, torch.autograd.grad(loss, [P[n] for n in names])))
    norms = np.array([float(grads[n].double().norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4)
    for n in SLICE_PARAMS:
        assert cosine(grads[n].flatten()[:256], g["gs_" + n]) > 0.99999, n
    r1, s1, c1 = O.hmr_forward(P, img[:1], init_pose=P["init_pose"] * 0.9, init_shape=P["init_shape"] + 0.1,
                               init_cam=P["init_cam"] * 1.1, n_iter=2)
    assert rel_err(r1.detach(), g["alt_rotmat"]) < 1e-5 and rel_err(c1.detach(), g["alt_cam"]) < 1e-5


def test_g4_losses(gmm_t, smpl_tabs):
    g = golden("g4_losses.npz")
    T = O.smpl_tables_to_torch(smpl_tabs)
    B = g["shape"].shape[0]
    rot = O.smplx_rodrigues(torch.from_numpy(g["aa"])).view(B, 24, 3, 3).requires_grad_(True)
    shape = torch.from_numpy(g["shape"]).requires_grad_(True)
    cam = torch.from_numpy(g["cam"]).requires_grad_(True)
    kp = torch.from_numpy(g["kp"])
    verts, j49 = O.smpl_forward(T, shape, rot[:, 1:], rot[:, 0:1], pose2rot=False)
    s2d = O.projection(cam, j49)
    assert rel_err(s2d.detach(), g["s2d"]) < 1e-5
    l2d, lsh, lpo = O.kp2d_loss(s2d, kp), O.shape_prior(shape), O.pose_prior(rot, gmm_t)
    assert abs(float(l2d) - g["l2d"]) < 1e-5 * abs(g["l2d"])
    assert abs(float(lsh) - g["lsh"]) < 1e-5 * abs(g["lsh"])
    assert abs(float(lpo) - g["lpo"]) < 1e-5 * abs(g["lpo"])
    loss = 10.0 * l2d + 2e-6 * lsh + 1e-4 * lpo
    gr, gs, gc = torch.autograd.grad(loss, [rot, shape, cam], retain_graph=True)
    assert rel_err(gr, g["g_rot"]) < 1e-4 and rel_err(gs, g["g_shape"]) < 1e-4 and rel_err(gc, g["g_cam"]) < 1e-4
    (gpo,) = torch.autograd.grad(lpo, rot)
    assert rel_err(gpo, g["gpo_rot"]) < 1e-4
    conf = kp[:, 25:, 2:3]
    l3d = O.s3d_loss(j49[:, 25:].detach(), torch.from_numpy(g["gt3"]), conf)
    assert abs(float(l3d) - g["l3d"]) < 1e-5 * abs(g["l3d"])


def test_g6_procrustes():
    g = golden("g6_procrustes.npz")
    hat = O.procrustes_align(g["S1"], g["S2"])
    np.testing.assert_allclose(hat, g["S1_hat"], rtol=1e-4, atol=1e-5)


# ---- third-party restatements: first-principles known answers (parity unpinned by reference) ----
def test_lbs_known_answers(smpl_tabs):
    T = O.smpl_tables_to_torch(smpl_tabs, torch.float64)
    g = torch.Generator().manual_seed(5)
    betas = torch.randn(2, 10, generator=g, dtype=torch.float64)
    eye = torch.eye(3, dtype=torch.float64).expand(2, 24, 3, 3).clone()
    verts, joints = O.lbs(T, betas, eye)
    v_shaped = T["v_template"][None] + torch.einsum("bl,vcl->bvc", betas, T["shapedirs"])
    assert torch.allclose(verts, v_shaped, atol=1e-12)                       # identity pose
    assert torch.allclose(joints, torch.einsum("jv,bvc->bjc", T["J_regressor"], v_shaped), atol=1e-12)
    # a global rotation of the root moves every vertex rigidly about joint 0
    R0 = O.smplx_rodrigues(torch.tensor([[0.4, -0.7, 0.3]], dtype=torch.float64))[0]
    rot = eye.clone()
    rot[:, 0] = R0
    verts2, _ = O.lbs(T, betas, rot)
    J0 = joints[:, :1]
    assert torch.allclose(verts2, (v_shaped - J0) @ R0.T + J0, atol=1e-10)   # posedirs see R[1:]-I = 0
    # fp32 agrees with fp64
    T32 = O.smpl_tables_to_torch(smpl_tabs, torch.float32)
    rr = O.smplx_rodrigues(torch.randn(48, 3, generator=g) * 0.3).view(2, 24, 3, 3)
    v32, _ = O.lbs(T32, betas.float(), rr)
    v64, _ = O.lbs(T, betas, rr.double())
    assert rel_err(v32, v64) < 1e-5


def test_maml_first_order_semantics():
    """FO outer gradient == gradient evaluated at the fast weights (identity through clone)."""
    torch.manual_seed(0)
    w = torch.randn(5, requires_grad=True)
    x = torch.randn(5)
    inner = lambda p: ((p * x).sum() - 1.0) ** 2
    outer = lambda p: (p ** 3).sum()
    fast = w.clone()
    (g,) = torch.autograd.grad(inner(fast), fast)
    fast = fast - 0.1 * g
    (go,) = torch.autograd.grad(outer(fast), w)
    wf = fast.detach().clone().requires_grad_(True)
    (gref,) = torch.autograd.grad(outer(wf), wf)
    assert torch.allclose(go, gref)
