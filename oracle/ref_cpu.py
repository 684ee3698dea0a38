"""CPU oracle for the DynaBOA per-frame adaptation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dynaboa_amd/`` imports this file; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may, and only as the
checker / the timed CPU baseline, never as the thing shipped.

It is a from-scratch fp32 PyTorch-CPU restatement of the reference algorithm; every function
cites the reference lines it follows.  Pinning (see tools/make_golden.py, tests/golden/):
  * everything whose arithmetic lives in /root/reference (HMR forward, geometry, GMM prior,
    loss assembly, the per-frame schedule, Procrustes) is checked against outputs of the
    reference's own code imported in the build container -> committed .npz goldens;
  * the two third-party pieces the reference only *calls* (smplx LBS, learn2learn MAML) are
    absent from /root/reference and not installable here.  They are restated from their
    published algorithms and pinned by first-principles known-answer
    tests only -> for those two, PARITY IS UNPINNED BY THE REFERENCE;
  * frame preprocessing: the box / keypoint arithmetic of utils/dataprocess.py is pinned by golden g7 (the reference's own
    crop() / transform() run here); the resize inside crop() is scikit-image 0.17.2's, not installed here: it is restated
    from its published defaults on scipy.ndimage (skimage_resize below) and g7 was generated with that restatement plugged
    into the reference's crop() -> the RESIZE is unpinned by the reference (known-answer tests only).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

FOCAL = 5000.0
IMG_RES = 224
BLOCKS = (3, 4, 6, 3)
GN_GROUPS = 4          # reference model/hmr.py:18  nn.GroupNorm(32 // 8, planes)
GN_EPS = 1e-5

Params = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------
# HMR  (reference model/hmr.py:40-60 Bottleneck, :127-181 HMR.forward)
# ----------------------------------------------------------------------------------------
def _gn(P: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.group_norm(x, GN_GROUPS, P[name + ".weight"], P[name + ".bias"], GN_EPS)


def _bottleneck(P: Params, pre: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    y = F.relu(_gn(P, pre + "bn1", F.conv2d(x, P[pre + "conv1.weight"])))
    y = F.relu(_gn(P, pre + "bn2", F.conv2d(y, P[pre + "conv2.weight"], stride=stride, padding=1)))
    y = _gn(P, pre + "bn3", F.conv2d(y, P[pre + "conv3.weight"]))
    if (pre + "downsample.0.weight") in P:
        x = _gn(P, pre + "downsample.1", F.conv2d(x, P[pre + "downsample.0.weight"], stride=stride))
    return F.relu(y + x)


def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """reference utils/geometry.py:47-61.  The six numbers are read as a (3,2) matrix, i.e.
    a1 = elements 0,2,4 and a2 = elements 1,3,5; b1,b2,b3 become the COLUMNS of R."""
    m = x.reshape(-1, 3, 2)
    a1, a2 = m[:, :, 0], m[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = a2 - (b1 * a2).sum(1, keepdim=True) * b1
    b2 = u / u.norm(dim=1, keepdim=True).clamp_min(1e-12)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=2)


def hmr_forward(P: Params, x: torch.Tensor, need_feature: bool = False,
                init_pose=None, init_shape=None, init_cam=None, n_iter: int = 3, return_pose6d: bool = False):
    """Functional HMR forward in eval mode (dropout = identity).  Returns
    (rotmat (B,24,3,3), shape (B,10), cam (B,3)[, 15 features])."""
    B = x.shape[0]
    pose = P["init_pose"].expand(B, -1) if init_pose is None else init_pose
    shape = P["init_shape"].expand(B, -1) if init_shape is None else init_shape
    cam = P["init_cam"].expand(B, -1) if init_cam is None else init_cam
    feats: List[torch.Tensor] = []
    x = F.conv2d(x, P["conv1.weight"], stride=2, padding=3)
    feats.append(x)
    x = F.max_pool2d(F.relu(_gn(P, "bn1", x)), 3, 2, 1)
    for li, nblk in enumerate(BLOCKS, start=1):
        for bi in range(nblk):
            x = _bottleneck(P, f"layer{li}.{bi}.", x, 2 if (bi == 0 and li > 1) else 1)
        feats.append(x)
    xf = F.avg_pool2d(x, 7, 1).flatten(1)
    feats.append(xf)
    for _ in range(n_iter):
        h = F.linear(torch.cat([xf, pose, shape, cam], 1), P["fc1.weight"], P["fc1.bias"])
        feats += [h.clone(), h.clone()]          # fc1 output, then the (identity) dropout of it
        h = F.linear(h, P["fc2.weight"], P["fc2.bias"])
        feats.append(h.clone())
        pose = F.linear(h, P["decpose.weight"], P["decpose.bias"]) + pose
        shape = F.linear(h, P["decshape.weight"], P["decshape.bias"]) + shape
        cam = F.linear(h, P["deccam.weight"], P["deccam.bias"]) + cam
    if return_pose6d:                 # the regressor's raw state (tests of the tangent / Hessian-vector passes)
        return pose, shape, cam
    R = rot6d_to_rotmat(pose).view(B, 24, 3, 3)
    return (R, shape, cam, feats) if need_feature else (R, shape, cam)


# ----------------------------------------------------------------------------------------
# Rotation conversions / projection (reference utils/geometry.py)
# ----------------------------------------------------------------------------------------
def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """reference utils/geometry.py:26-45 (w,x,y,z), normalised first."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    rows = [w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
            2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
            2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]
    return torch.stack(rows, 1).view(-1, 3, 3)


def batch_rodrigues(theta: torch.Tensor) -> torch.Tensor:
    """reference utils/geometry.py:9-24: angle = ||theta + 1e-8||, half-angle quaternion."""
    ang = (theta + 1e-8).norm(dim=1, keepdim=True)
    axis = theta / ang
    half = 0.5 * ang
    return quat_to_rotmat(torch.cat([half.cos(), half.sin() * axis], 1))


def rotmat_to_quat(R: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """reference utils/geometry.py:248-306 (kornia-derived 4-branch formula).  The reference
    transposes first; written here directly on R (m_ij of the reference = R_ji)."""
    r00, r01, r02 = R[:, 0, 0], R[:, 0, 1], R[:, 0, 2]
    r10, r11, r12 = R[:, 1, 0], R[:, 1, 1], R[:, 1, 2]
    r20, r21, r22 = R[:, 2, 0], R[:, 2, 1], R[:, 2, 2]
    d2 = r22 < eps
    d01 = r00 > r11
    d0n1 = r00 < -r11
    t0 = 1 + r00 - r11 - r22
    q0 = torch.stack([r21 - r12, t0, r10 + r01, r02 + r20], -1)
    t1 = 1 - r00 + r11 - r22
    q1 = torch.stack([r02 - r20, r10 + r01, t1, r21 + r12], -1)
    t2 = 1 - r00 - r11 + r22
    q2 = torch.stack([r10 - r01, r02 + r20, r21 + r12, t2], -1)
    t3 = 1 + r00 + r11 + r22
    q3 = torch.stack([t3, r21 - r12, r02 - r20, r10 - r01], -1)
    c0 = (d2 & d01).unsqueeze(1).type_as(R)
    c1 = (d2 & ~d01).unsqueeze(1).type_as(R)
    c2 = (~d2 & d0n1).unsqueeze(1).type_as(R)
    c3 = (~d2 & ~d0n1).unsqueeze(1).type_as(R)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    t = t0[:, None] * c0 + t1[:, None] * c1 + t2[:, None] * c2 + t3[:, None] * c3
    return 0.5 * q / t.sqrt()


def quat_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    """reference utils/geometry.py:216-245."""
    w, v = q[:, 0], q[:, 1:]
    s2 = (v * v).sum(1)
    s = s2.sqrt()
    two_theta = 2.0 * torch.where(w < 0, torch.atan2(-s, -w), torch.atan2(s, w))
    k = torch.where(s2 > 0, two_theta / s, torch.full_like(s, 2.0))
    return v * k[:, None]


def rotmat_to_axis_angle(R: torch.Tensor) -> torch.Tensor:
    """reference utils/geometry.py:184-213 (3x3 input branch; NaN -> 0)."""
    aa = quat_to_axis_angle(rotmat_to_quat(R.reshape(-1, 3, 3)))
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def projection(cam: torch.Tensor, s3d: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
    """reference base_adaptor.py:160-170 + utils/geometry.py:63-91 with R=I, centre 0:
    returns the [-1,1]-normalised 2-D points ('normed')."""
    t = torch.stack([cam[:, 1], cam[:, 2], 2 * FOCAL / (IMG_RES * cam[:, 0] + eps)], -1)
    p = s3d + t[:, None, :]
    p = p / p[:, :, 2:3]
    return (FOCAL * p[:, :, :2]) / (IMG_RES / 2.0)


# ----------------------------------------------------------------------------------------
# SMPL linear blend skinning  [third-party smplx; SURVEY Appendix B]  PARITY UNPINNED
# ----------------------------------------------------------------------------------------
def smplx_rodrigues(rv: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1-cos) K^2."""
    ang = (rv + 1e-8).norm(dim=1, keepdim=True)
    k = rv / ang
    z = torch.zeros_like(k[:, 0])
    K = torch.stack([z, -k[:, 2], k[:, 1], k[:, 2], z, -k[:, 0], -k[:, 1], k[:, 0], z], 1).view(-1, 3, 3)
    s, c = ang.sin()[:, :, None], ang.cos()[:, :, None]
    return torch.eye(3, dtype=rv.dtype) + s * K + (1 - c) * (K @ K)


def lbs(T: Dict[str, torch.Tensor], betas: torch.Tensor, rot: torch.Tensor):
    """betas (B,10), rot (B,24,3,3) -> verts (B,6890,3), posed joints (B,24,3)."""
    B = betas.shape[0]
    v_shaped = T["v_template"][None] + torch.einsum("bl,vcl->bvc", betas, T["shapedirs"])
    J = torch.einsum("jv,bvc->bjc", T["J_regressor"], v_shaped)
    pf = (rot[:, 1:] - torch.eye(3, dtype=rot.dtype)).reshape(B, -1)
    v_posed = v_shaped + (pf @ T["posedirs"]).view(B, -1, 3)
    parents = T["parents"].tolist()
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    GR: List[torch.Tensor] = [rot[:, 0]]
    Gt: List[torch.Tensor] = [rel[:, 0]]
    for i in range(1, 24):
        p = parents[i]
        GR.append(GR[p] @ rot[:, i])
        Gt.append((GR[p] @ rel[:, i, :, None])[..., 0] + Gt[p])
    GRs, Gts = torch.stack(GR, 1), torch.stack(Gt, 1)                # (B,24,3,3), (B,24,3)
    At = Gts - (GRs @ J[..., None])[..., 0]
    W = T["lbs_weights"]
    TR = torch.einsum("vj,bjrc->bvrc", W, GRs)
    Tt = torch.einsum("vj,bjr->bvr", W, At)
    verts = (TR @ v_posed[..., None])[..., 0] + Tt
    return verts, Gts


def smpl_forward(T: Dict[str, torch.Tensor], betas: torch.Tensor, body_pose: torch.Tensor,
                 global_orient: torch.Tensor, pose2rot: bool = True):
    """reference model/smpl.py:25-37 on top of smplx.SMPL.forward: 24 posed joints + 21
    selected vertices + 9 regressed extras, gathered to 49 joints.  Returns (verts, joints49)."""
    B = betas.shape[0]
    if pose2rot:
        full = torch.cat([global_orient.reshape(B, -1, 3), body_pose.reshape(B, -1, 3)], 1)
        rot = smplx_rodrigues(full.reshape(-1, 3)).view(B, 24, 3, 3)
    else:
        rot = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, 23, 3, 3)], 1)
    verts, jt = lbs(T, betas, rot)
    joints = torch.cat([jt, verts[:, T["vertex_joint_ids"]],
                        torch.einsum("ev,bvc->bec", T["J_regressor_extra"], verts)], 1)
    return verts, joints[:, T["joint_map"]]


def smpl_tables_to_torch(tab: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    from dynaboa_amd import constants as C   # constants only (no compute)
    out = {k: torch.as_tensor(tab[k]).to(dtype) for k in
           ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J_regressor_extra")}
    if "J_regressor_h36m" in tab:
        out["J_regressor_h36m"] = torch.as_tensor(tab["J_regressor_h36m"]).to(dtype)
    out["parents"] = torch.as_tensor(tab["parents"]).long()
    out["vertex_joint_ids"] = torch.tensor(C.VERTEX_JOINT_IDS, dtype=torch.long)
    out["joint_map"] = torch.tensor(C.JOINT_MAP_49, dtype=torch.long)
    return out


# ----------------------------------------------------------------------------------------
# Priors and losses (reference base_adaptor.py:222-422, utils/smplify/prior.py:181-196)
# ----------------------------------------------------------------------------------------
def gmm_prior(pose69: torch.Tensor, gmm: Dict[str, torch.Tensor]) -> torch.Tensor:
    d = pose69[:, None, :] - gmm["means"][None]
    quad = (torch.einsum("mij,bmj->bmi", gmm["precisions"], d) * d).sum(-1)
    return (0.5 * quad - gmm["nll_weights"].log()).min(dim=1).values


def pose_prior(rotmat: torch.Tensor, gmm) -> torch.Tensor:
    """reference base_adaptor.py:405-409."""
    aa = rotmat_to_axis_angle(rotmat[:, 1:].reshape(-1, 3, 3)).reshape(-1, 69)
    return gmm_prior(aa, gmm).mean()


def shape_prior(betas: torch.Tensor) -> torch.Tensor:
    """reference base_adaptor.py:401-402."""
    return (betas ** 2).sum(-1).mean()


def kp2d_loss(pred_s2d: torch.Tensor, gt_kp: torch.Tensor) -> torch.Tensor:
    """reference base_adaptor.py:229,234: confidence-masked MSE over the 24 GT-style joints."""
    conf = gt_kp[:, 25:, 2:3]
    return (((pred_s2d[:, 25:] - gt_kp[:, 25:, :2]) ** 2) * conf).mean()


def s3d_loss(pred_s3d24, gt_s3d24, conf):
    """reference base_adaptor.py:412-422 (hip-centred, masked)."""
    g = gt_s3d24 - ((gt_s3d24[:, 2] + gt_s3d24[:, 3]) / 2)[:, None]
    p = pred_s3d24 - ((pred_s3d24[:, 2] + pred_s3d24[:, 3]) / 2)[:, None]
    return (conf * (p - g) ** 2).mean()


DEFAULT_OPTS = dict(lr=3e-6, beta1=0.5, beta2=0.9, fastlr=8e-6, inner_step=1,
                    s2dloss_weight=10.0, shape_prior_weight=2e-6, pose_prior_weight=1e-4,
                    labelloss_weight=0.1, teacherloss_weight=0.1, alpha=0.1,
                    motionloss_weight=0.8, interval=5, cos_sim_threshold=3.1e-4, optim_steps=7,
                    use_frame_losses_lower=1, use_frame_losses_upper=1,
                    use_temporal_losses_lower=0, use_temporal_losses_upper=1,
                    lower_level_mixtrain=1, upper_level_mixtrain=1, retrieval=1,
                    use_meanteacher=1, use_motion=1, dynamic_boa=1, sample_num=1)


class Adapter:
    """State + loss assembly + the per-frame bilevel schedule, first-order MAML.

    Follows reference base_adaptor.py:222-398 (level losses) and dynaboa_benchmark.py:126-193
    (schedule).  MAML clone/adapt restated from learn2learn 0.1.5:
    clone = per-parameter ``p.clone()``; adapt = ``p' = p - fastlr * dL/dp`` with
    ``create_graph = not first_order``; FO outer gradient = dL_up/dtheta' (identity through
    the clone).  Adam restated from torch.optim.Adam (no amsgrad / weight decay).
    """

    def __init__(self, P: Params, smpl: Dict[str, torch.Tensor], gmm: Dict[str, torch.Tensor],
                 opts: Optional[dict] = None, first_order: bool = True):
        self.o = dict(DEFAULT_OPTS)
        self.o.update(opts or {})
        self.buf = {k: v.clone() for k, v in P.items() if k.startswith("init_")}
        self.theta = {k: v.clone().requires_grad_(True) for k, v in P.items()
                      if not k.startswith("init_")}
        self.teacher = {k: v.detach().clone() for k, v in P.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.theta.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.theta.items()}
        self.adam_t = 0
        self.smpl, self.gmm = smpl, gmm
        self.first_order = first_order
        self.history: Dict[int, dict] = {}
        self.global_step = 0
        self.log: Dict[str, float] = {}
        self.exemplar_fn = None          # step -> exemplar batch (stands in for retrieval())

    # -- helpers -------------------------------------------------------------------------
    def _full(self, w: Params) -> Params:
        d = dict(w)
        d.update(self.buf)
        return d

    def decode(self, rotmat, shape):
        verts, j49 = smpl_forward(self.smpl, shape, rotmat[:, 1:], rotmat[:, 0:1], pose2rot=False)
        return j49, verts

    def frame_losses(self, rotmat, shape, s2d, kp, tag):
        l2d = kp2d_loss(s2d, kp)
        lsh = shape_prior(shape)
        lpo = pose_prior(rotmat, self.gmm)
        self.log[f"{tag}/s2dloss"], self.log[f"{tag}/shape_prior"], self.log[f"{tag}/pose_prior"] = \
            float(l2d), float(lsh), float(lpo)
        return (l2d * self.o["s2dloss_weight"] + lsh * self.o["shape_prior_weight"]
                + lpo * self.o["pose_prior_weight"])

    def teacher_loss(self, image, rotmat, shape, s2d, s3d):
        """reference base_adaptor.py:320-343."""
        with torch.no_grad():
            tr, ts, tc = hmr_forward(self.teacher, image)
            t3d, _ = self.decode(tr, ts)
            t2d = projection(tc, t3d)
        return (5 * F.mse_loss(s2d, t2d) + 5 * F.mse_loss(t3d, s3d)
                + 1e-3 * F.mse_loss(shape, ts) + F.mse_loss(rotmat, tr))

    def motion_loss(self, w, s2d24, kp24):
        """reference base_adaptor.py:379-398."""
        h = self.history[self.global_step - self.o["interval"]]
        hr, hs, hc = hmr_forward(self._full(w), h["image"])
        h3d, _ = self.decode(hr, hs)
        h2d = projection(hc, h3d)
        pm = s2d24 - h2d[:, 25:]
        gm = kp24[:, :, :2] - h["s2d"][:, 25:, :2]
        conf = ((h["s2d"][:, 25:, 2:3] + kp24[:, :, 2:3]) == 2).float()
        return (((pm - gm) ** 2) * conf).mean()

    def label_loss(self, w, ex):
        """reference base_adaptor.py:346-376."""
        conf = ex["keypoints"][:, 25:, 2:3]
        r, s, c = hmr_forward(self._full(w), ex["img"])
        j3d, _ = self.decode(r, s)
        gt_r = batch_rodrigues(ex["pose"].reshape(-1, 3)).view(-1, 24, 3, 3)
        s2d = projection(c, j3d)
        l2d = (((s2d[:, 25:] - ex["keypoints"][:, 25:, :2]) ** 2) * conf).mean()
        l3d = s3d_loss(j3d[:, 25:], ex["pose_3d"][:, :, :3], conf)
        return 5 * l2d + 5 * l3d + 1e-3 * F.mse_loss(s, ex["betas"]) + F.mse_loss(r, gt_r)

    def level_loss(self, w, image, kp, level: str):
        """lower/upper_level_adaptation (reference base_adaptor.py:222-317)."""
        o = self.o
        r, s, c, feats = hmr_forward(self._full(w), image, need_feature=True)
        s3d, _ = self.decode(r, s)
        s2d = projection(c, s3d)
        loss = None
        if o[f"use_frame_losses_{level}"]:
            loss = self.frame_losses(r, s, s2d, kp, "ll" if level == "lower" else "ul")
        if o[f"use_temporal_losses_{level}"]:
            if o["use_meanteacher"]:
                tl = self.teacher_loss(image, r, s, s2d, s3d) * o["teacherloss_weight"]
                loss = tl if loss is None else loss + tl
            if o["use_motion"] and (self.global_step - o["interval"]) > 0:
                loss = loss + self.motion_loss(w, s2d[:, 25:], kp[:, 25:]) * o["motionloss_weight"]
        if o[f"{level}_level_mixtrain"]:
            ex = self.exemplar_fn(self.global_step)
            loss = loss + self.label_loss(w, ex) * o["labelloss_weight"]
        self.log[("ll" if level == "lower" else "ul") + "/total"] = float(loss)
        return loss, feats

    def adam_step(self, grads: Params):
        """torch.optim.Adam single-tensor formula."""
        o = self.o
        self.adam_t += 1
        b1, b2, t = o["beta1"], o["beta2"], self.adam_t
        bc1, bc2s = 1 - b1 ** t, (1 - b2 ** t) ** 0.5
        with torch.no_grad():
            for k, p in self.theta.items():
                g = grads[k]
                self.m[k].lerp_(g, 1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                p.addcdiv_(self.m[k], (self.v[k].sqrt() / bc2s).add_(1e-8), value=-o["lr"] / bc1)

    def ema_teacher(self):
        a = self.o["alpha"]
        with torch.no_grad():
            for k, p in self.theta.items():
                self.teacher[k].mul_(a).add_(p, alpha=1 - a)

    def predict(self, w, image):
        with torch.no_grad():
            r, s, c = hmr_forward(self._full(w), image)
            j49, verts = self.decode(r, s)
        return dict(rotmat=r, shape=s, cam=c, joints=j49, verts=verts)

    # -- the per-frame schedule ------------------------------------------------------------
    def adapt_frame(self, batch: Dict[str, torch.Tensor]) -> dict:
        """dynaboa_benchmark.py:126-193 (use_boa branch).  Returns a record of what happened
        (losses, the outer gradient, predictions) for parity checks."""
        o = self.o
        image, kp = batch["image"], batch["smpl_j2d"]
        self.history[self.global_step] = dict(image=image.clone(), s2d=kp.clone())
        rec: dict = dict(lower_loss=[], upper_loss=[], extra_steps=0)
        with torch.no_grad():
            init_feats = hmr_forward(self._full(self.theta), image, need_feature=True)[3]
        so = not self.first_order
        fast = {k: p.clone() for k, p in self.theta.items()}
        for _ in range(o["inner_step"]):
            loss, _ = self.level_loss(fast, image, kp, "lower")
            names = list(fast)
            g = torch.autograd.grad(loss, [fast[n] for n in names], retain_graph=so, create_graph=so)
            fast = {n: fast[n] - o["fastlr"] * gi for n, gi in zip(names, g)}
            rec["lower_loss"].append(float(loss))
        loss, _ = self.level_loss(fast, image, kp, "upper")
        names = list(self.theta)
        g = torch.autograd.grad(loss, [self.theta[n] for n in names])
        grads = dict(zip(names, g))
        rec["upper_loss"].append(float(loss))
        rec["outer_grad"] = grads
        self.adam_step(grads)
        if o["use_meanteacher"]:
            self.ema_teacher()
        if o["dynamic_boa"]:
            with torch.no_grad():
                feats = hmr_forward(self._full(self.theta), image, need_feature=True)[3]
            cos = float(F.cosine_similarity(init_feats[12].flatten(), feats[12].flatten(), dim=0, eps=1e-12))
            rec["gate_cos12"] = [cos]                                  # every check of the gate (reference: self.feat_sims[step][k][12])
            steps = 0
            while 1 - cos > o["cos_sim_threshold"]:
                steps += 1
                if steps > o["optim_steps"]:
                    break
                loss, prev = self.level_loss(self.theta, image, kp, "upper")
                g = torch.autograd.grad(loss, [self.theta[n] for n in names])
                self.adam_step(dict(zip(names, g)))
                rec["upper_loss"].append(float(loss))
                if o["use_meanteacher"]:
                    self.ema_teacher()
                with torch.no_grad():
                    feats = hmr_forward(self._full(self.theta), image, need_feature=True)[3]
                cos = float(F.cosine_similarity(prev[12].detach().flatten(), feats[12].flatten(), dim=0, eps=1e-12))
                rec["gate_cos12"].append(cos)
            rec["extra_steps"] = steps
        rec["pred"] = self.predict(self.theta, image)
        self.global_step += 1
        return rec


# ----------------------------------------------------------------------------------------
# Metrics (reference utils/pose_utils.py:9-64, dynaboa_benchmark.py:204-262)
# ----------------------------------------------------------------------------------------
def procrustes_align(S1: np.ndarray, S2: np.ndarray) -> np.ndarray:
    """Similarity-align each (N,3) set S1[i] to S2[i]; returns aligned S1."""
    out = np.zeros_like(S1)
    for i in range(S1.shape[0]):
        a, b = S1[i].T, S2[i].T
        mu1, mu2 = a.mean(1, keepdims=True), b.mean(1, keepdims=True)
        X1, X2 = a - mu1, b - mu2
        K = X1 @ X2.T
        U, _, Vh = np.linalg.svd(K)
        V = Vh.T
        Z = np.eye(3)
        Z[-1, -1] = np.sign(np.linalg.det(U @ V.T))
        R = V @ Z @ U.T
        scale = np.trace(R @ K) / (X1 ** 2).sum()
        out[i] = (scale * R @ a + (mu2 - scale * R @ mu1)).T
    return out


def eval_metrics(pred_verts, gt_verts_gendered, gt_verts_neutral, J_h36m, j14: Sequence[int]):
    """MPJPE / PA-MPJPE / PVE in mm (reference dynaboa_benchmark.py:220-262)."""
    def j(v):
        k = torch.einsum("jv,bvc->bjc", J_h36m, v)
        return k[:, list(j14)] - k[:, :1]
    p, g = j(pred_verts), j(gt_verts_gendered)
    mpjpe = (p - g).norm(dim=-1).mean(-1).numpy()
    pa = np.sqrt(((procrustes_align(p.numpy(), g.numpy()) - g.numpy()) ** 2).sum(-1)).mean(-1)
    pve = float((gt_verts_neutral - pred_verts).norm(dim=-1).mean())
    return mpjpe * 1000, pa * 1000, pve * 1000


# ----------------------------------------------------------------------------------------
# Frame preprocessing (reference utils/dataprocess.py:12-96, boa_dataset/pw3d.py:117-156, base_adaptor.py:510-533)
# ----------------------------------------------------------------------------------------
def get_transform(center, scale, res):
    """utils/dataprocess.py:12-37 with rot = 0."""
    h = 200 * scale
    t = np.zeros((3, 3))
    t[0, 0] = float(res[1]) / h
    t[1, 1] = float(res[0]) / h
    t[0, 2] = res[1] * (-float(center[0]) / h + .5)
    t[1, 2] = res[0] * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    return t


def transform_pt(pt, center, scale, res, invert=0):
    """utils/dataprocess.py:39-46: pixel location to / from the crop frame (1-based in, truncation, 1-based out)."""
    t = get_transform(center, scale, res)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.dot(t, np.array([pt[0] - 1, pt[1] - 1, 1.]).T)
    return new_pt[:2].astype(int) + 1


def crop_box(center, scale, res):
    """Upper-left / bottom-right corner of the crop box in frame pixels (utils/dataprocess.py:51-54)."""
    ul = np.array(transform_pt([1, 1], center, scale, res, invert=1)) - 1
    br = np.array(transform_pt([res[0] + 1, res[1] + 1], center, scale, res, invert=1)) - 1
    return ul, br


def skimage_resize(image: np.ndarray, output_shape) -> np.ndarray:
    """skimage.transform.resize(image, output_shape) of scikit-image 0.17.2 (reference requirements.txt:19) with its
    defaults, restated - the package is not installed here, so THIS restatement is unpinned by the reference: order 1,
    mode 'reflect' (scipy 'mirror'), anti_aliasing on: ndimage.gaussian_filter with sigma = max(0, (in/out - 1)/2) per axis
    (0 along channels), then warp() by the axis-aligned affine map  in = (out + 0.5) * in/out - 0.5  with bilinear
    interpolation (floor / ceil neighbours, mirror boundary)."""
    from scipy import ndimage as ndi
    image = np.asarray(image, np.float64)
    out_shape = tuple(output_shape) + (image.shape[-1],) if len(output_shape) == image.ndim - 1 else tuple(output_shape)
    factors = np.asarray(image.shape, float) / np.asarray(out_shape, float)
    sigma = np.maximum(0, (factors - 1) / 2)
    image = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")

    def mirror(i, n):
        if n == 1:
            return np.zeros_like(i)
        p = 2 * (n - 1)
        i = np.mod(i, p)
        return np.where(i < n, i, p - i)
    rows, cols = out_shape[0], out_shape[1]
    r = factors[0] * (np.arange(rows) + 0.5) - 0.5
    c = factors[1] * (np.arange(cols) + 0.5) - 0.5
    r0, r1 = np.floor(r).astype(int), np.ceil(r).astype(int)
    c0, c1 = np.floor(c).astype(int), np.ceil(c).astype(int)
    dr, dc = (r - r0)[:, None, None], (c - c0)[None, :, None]
    R0, R1, C0, C1 = mirror(r0, image.shape[0]), mirror(r1, image.shape[0]), mirror(c0, image.shape[1]), mirror(c1, image.shape[1])
    top = (1 - dc) * image[R0][:, C0] + dc * image[R0][:, C1]
    bot = (1 - dc) * image[R1][:, C0] + dc * image[R1][:, C1]
    return (1 - dr) * top + dr * bot


def crop(img: np.ndarray, center, scale, res) -> np.ndarray:
    """utils/dataprocess.py:48-96 with rot = 0: paste the box (zero outside the frame), resize to `res`."""
    ul, br = crop_box(center, scale, res)
    new_shape = [br[1] - ul[1], br[0] - ul[0], img.shape[2]]
    new_img = np.zeros(new_shape)
    new_x = max(0, -ul[0]), min(br[0], len(img[0])) - ul[0]
    new_y = max(0, -ul[1]), min(br[1], len(img)) - ul[1]
    old_x = max(0, ul[0]), min(len(img[0]), br[0])
    old_y = max(0, ul[1]), min(len(img), br[1])
    new_img[new_y[0]:new_y[1], new_x[0]:new_x[1]] = img[old_y[0]:old_y[1], old_x[0]:old_x[1]]
    return skimage_resize(new_img, res)


IMG_NORM_MEAN, IMG_NORM_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # reference constants.py:15-16


def rgb_processing(rgb_img: np.ndarray, center, scale, res=224) -> np.ndarray:
    """boa_dataset/pw3d.py:131-136 + :121-123 (test time): crop, HWC->CHW, /255, Normalize.  -> (3, res, res) float32."""
    x = crop(rgb_img.copy(), center, scale, [res, res])
    x = np.transpose(x.astype("float32"), (2, 0, 1)) / 255.0
    return ((x - np.array(IMG_NORM_MEAN, np.float32)[:, None, None]) / np.array(IMG_NORM_STD, np.float32)[:, None, None]).astype(np.float32)


def j2d_processing(kp: np.ndarray, center, scale, res=224) -> np.ndarray:
    """boa_dataset/pw3d.py:138-151 (test time: no flip): keypoints into the crop frame, then to [-1, 1]."""
    kp = kp.copy()
    for i in range(kp.shape[0]):
        kp[i, 0:2] = transform_pt(kp[i, 0:2] + 1, center, scale, [res, res])
    kp[:, :-1] = 2. * kp[:, :-1] / res - 1.
    return kp.astype("float32")
