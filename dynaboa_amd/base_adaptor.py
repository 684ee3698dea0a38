"""BaseAdaptor: model / optimiser / teacher set-up, loss assembly for the lower and upper level of
the bilevel adaptation, mean-teacher EMA, frame history and exemplar retrieval.

Mirrors the public surface of reference ``base_adaptor.py::BaseAdaptor`` (lines 36-447): same
constructor argument (the argparse namespace), same method names and return conventions
(``projection`` -> {'ori','normed'}, ``decode_smpl_params`` -> {'s3d','vts'},
``lower/upper_level_adaptation(image, gt_keypoints_2d, h36m_batch, learner)`` -> (loss, features),
``cal_*``), same attributes the drivers read.  What differs is underneath: the model is the native
HIP engine, SMPL / losses / Adam / EMA are HIP kernels, and the frame history stays on the device
instead of round-tripping through NumPy (reference :173-180).

Data that the reference reads from fixed paths (checkpoint, SMPL pickles, regressors, retrieval
clusters) can be injected through ``assets_bundle`` so the path runs on synthetic stand-ins
(dynaboa_amd.assets) where the licensed files are absent.
"""
from __future__ import annotations

import os
import random
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import assets as A
from . import constants
from .hmr import hmr
from .fused_level import clear_last_forward, level_forward
from .losses import (AUX_MAX_BATCH, MaxMixturePrior, frame_losses, labelled_term, motion_term, pose_prior, projection_normed,
                     teacher_term)
from .maml import MAML
from .optim import Adam, ema_update
from .smpl import SMPL


def synthetic_bundle(seed: int = 22, identity_pose: bool = True, randomize_norm: bool = False, smpl_seed: int = 0,
                     resident_exemplars: bool = False):
    """Everything BaseAdaptor needs, generated from seeds."""
    mp = A.make_smpl_mean_params(identity_pose=identity_pose, seed=3)
    return SimpleNamespace(
        mean_params=mp,
        checkpoint=A.make_synthetic_checkpoint(seed, mp, randomize_norm=randomize_norm),
        smpl_neutral=A.make_synthetic_smpl(smpl_seed), smpl_male=A.make_synthetic_smpl(smpl_seed + 1),
        smpl_female=A.make_synthetic_smpl(smpl_seed + 2),
        exemplars=lambda step, n: A.make_exemplars_pinned(step, n),
        # opt-in: exemplars resident on the device and shared by every sequence of a step; off = every retrieval()
        # uploads its own copy, as the reference does (base_adaptor.py:82-96)
        exemplars_device=(lambda step, n, device: A.make_exemplars_device(step, n, device)) if resident_exemplars else None,
        dataloader=None, gmm_folder=None)


class BaseAdaptor:
    def __init__(self, options, assets_bundle=None, device=None):
        self.options = options
        self.exppath = os.path.join(getattr(options, "expdir", "exps"), getattr(options, "expname", "run"))
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.bundle = assets_bundle
        self.seed_everything(options.seed)
        self.options.mixtrain = options.lower_level_mixtrain or options.upper_level_mixtrain
        self.history: Dict[int, dict] = {}
        self.fit_losses: Dict[str, torch.Tensor] = {}
        self.kp2dlosses_lower, self.kp2dlosses_upper = [], {}
        self.global_step = 0
        if options.retrieval:
            self.load_h36_cluster_res()
            if self.bundle is None:
                # the exemplar set retrieval() draws from (reference base_adaptor.py:55)
                from . import datasets as D
                self.h36m_dataset = D.SourceDataset('data/retrieval_res/h36m_random_sample_center_10_10.pt',
                                                    img_dir=getattr(options, "h36m_root", None) or D.H36M_ROOT, device=self.device)
        self.set_model_optim()
        if options.use_meanteacher:
            self.set_teacher()
        self.set_dataloader()
        self.set_criterion()
        self.setup_smpl()

    # ------------------------------------------------------------------ set-up
    def seed_everything(self, seed):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)

    def _checkpoint(self):
        if self.bundle is not None:
            return self.bundle.checkpoint
        return torch.load(self.options.model_file, map_location="cpu")

    def _mean_params(self):
        return self.bundle.mean_params if self.bundle is not None else "data/smpl_mean_params.npz"

    def set_model_optim(self):
        ck = self._checkpoint()["model"]
        model = hmr(self._mean_params(), seed=0)
        if self.options.use_boa:
            # the reference hard-codes first_order=True (base_adaptor.py:119); `second_order` is this build's switch
            self.model = MAML(model, lr=self.options.fastlr,
                              first_order=not getattr(self.options, "second_order", 0)).to(self.device)
            self.model.load_state_dict(ck, strict=True)
        else:
            self.model = model.to(self.device)
            self.model.load_state_dict({k.replace("module.", ""): v for k, v in ck.items()}, strict=True)
        self.optimizer = Adam(self.model.parameters(), lr=self.options.lr, betas=(self.options.beta1, self.options.beta2))
        if self.options.use_boa and getattr(self.options, "second_order", 0):
            # this optimiser understands a gradient left as (v, H v, lr): one fused "Adam + accumulate" launch for the outer step
            self.model.defer_accumulate = bool(getattr(self.options, "fused_so_adam", 1))
        # precision of the backbone convolutions is a property of the engine plan (one per batch size, shared by every
        # model of the process): set explicitly both ways so that a bf16 run does not leak into a later fp32 one
        from .hmr import get_layout
        want = bool(getattr(self.options, "bf16_mfma", 0))
        for b in {1, int(getattr(self.options, "batch_size", 1)), int(getattr(self.options, "sample_num", 1))}:
            get_layout(b).set_bf16(want)

    def set_teacher(self):
        teacher = hmr(self._mean_params(), seed=0)
        for p in teacher.parameters():
            p.detach_()
        self.teacher = teacher.to(self.device)
        ck = self._checkpoint()["model"]
        self.teacher.load_state_dict({k.replace("module.", ""): v for k, v in ck.items()}, strict=True)
        # The reference never calls teacher.eval() (base_adaptor.py:151-158): its teacher runs with live Dropout in the
        # regressor, so the teacher targets carry p = 0.5 mask noise.  teacher_dropout=1 reproduces that (the masks come from
        # this build's counter-based generator, so runs agree with the reference in distribution, not sample by sample);
        # the default keeps the teacher deterministic (eval), which is what the goldens pin.  Recorded in DESIGN.md.
        if getattr(self.options, "teacher_dropout", 0):
            self.teacher.train()
        else:
            self.teacher.eval()

    def set_dataloader(self):
        """reference base_adaptor.py:130-137: the 3DPW test stream in sequence order, batch_size frames at a time, decoded
        ahead on 8 host threads; crop / resize / normalise on the GPU (datasets.py).  With a synthetic bundle the stream
        is whatever the caller iterates."""
        if self.bundle is not None:
            self.dataloader = self.bundle.dataloader
            return
        from . import datasets as D
        if getattr(self.options, "dataset", "3dpw") != "3dpw":
            raise NotImplementedError("the 'internet' demo dataset (reference boa_dataset/internet_data.py) is out of scope "
                                      "(SURVEY 2); pass frames to excute() yourself")
        self.imgdir = getattr(self.options, "pw3d_root", None) or D.PW3D_ROOT
        ds = D.PW3D(self.options, img_dir=self.imgdir, device=self.device)
        self.dataloader = D.FrameLoader(ds, batch_size=self.options.batch_size, workers=8) if len(ds) else None

    def set_criterion(self):
        folder = self.bundle.gmm_folder if self.bundle is not None else "data/spin_data"
        self.gmm_f = MaxMixturePrior(prior_folder=folder, num_gaussians=8).to(self.device)

    def setup_smpl(self):
        if self.bundle is not None:
            self.smpl_neutral = SMPL(tables=self.bundle.smpl_neutral).to(self.device)
            self.smpl_male = SMPL(tables=self.bundle.smpl_male).to(self.device)
            self.smpl_female = SMPL(tables=self.bundle.smpl_female).to(self.device)
            self.J_regressor = torch.from_numpy(self.bundle.smpl_neutral["J_regressor_h36m"]).float()
        else:
            self.smpl_neutral = SMPL("data/smpl", create_transl=False).to(self.device)
            self.smpl_male = SMPL("data/smpl", gender="male", create_transl=False).to(self.device)
            self.smpl_female = SMPL("data/smpl", gender="female", create_transl=False).to(self.device)
            self.J_regressor = torch.from_numpy(np.load("data/J_regressor_h36m.npy")).float()
        self.joint_mapper_h36m = list(constants.H36M_TO_J14)
        self.joint_mapper_gt = list(constants.J24_TO_J14)

    # ------------------------------------------------------------------ retrieval (base_adaptor.py:74-96)
    def load_h36_cluster_res(self):
        """reference base_adaptor.py:74-80: cluster centres of the base model's Human3.6M features + member indices."""
        self.centers = None
        if self.bundle is None:
            import joblib
            res = self.h36m_cluster_res = joblib.load("data/retrieval_res/cluster_res_random_sample_center_10_10_potocol2.pt")
            self.centers = torch.from_numpy(np.asarray(res["centers"])).float().to(self.device)
            self.index = res["index"]

    def get_h36m_data(self, indice):
        return dict(self.h36m_dataset[indice])

    def retrieval(self, feature):
        """reference base_adaptor.py:82-96: nearest cluster by cosine distance of features[5] (the pooled 2048-vector) to
        the centres, `sample_num` members drawn with the seeded `random` module, their items concatenated along dim 0.
        The cluster index is the one host synchronisation of the level (the reference's `.item()`)."""
        if self.bundle is not None:
            dev_fn = getattr(self.bundle, "exemplars_device", None)
            if dev_fn is not None:            # synthetic exemplars: resident on the device, one generation per step for all sequences
                return dev_fn(self.global_step, self.options.sample_num, self.device)
            batch = self.bundle.exemplars(self.global_step, self.options.sample_num)
        else:
            dists = 1 - F.cosine_similarity(feature, self.centers)
            pos_cluster = torch.argsort(dists)[0].item()
            pos_indices = random.sample(self.index[pos_cluster], self.options.sample_num)
            items = [self.get_h36m_data(i) for i in pos_indices]
            batch = items[0]
            for it in items[1:]:
                for k, v in it.items():
                    batch[k] = torch.cat([batch[k], v], dim=0) if torch.is_tensor(v) else batch[k]
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    # ------------------------------------------------------------------ geometry helpers
    def projection(self, cam, s3d, eps=1e-9):
        normed = projection_normed(cam, s3d)
        return {"ori": normed * (constants.IMG_RES / 2.0), "normed": normed}

    def decode_smpl_params(self, poses, beta, gender="neutral", pose2rot=False):
        smpl = {"neutral": self.smpl_neutral, "male": self.smpl_male, "female": self.smpl_female}[gender]
        out = smpl(betas=beta, body_pose=poses[:, 1:], global_orient=poses[:, 0].unsqueeze(1), pose2rot=pose2rot)
        return {"s3d": out.joints, "vts": out.vertices}

    # ------------------------------------------------------------------ history (device resident)
    def save_hist(self, image, s2d):
        self.history[self.global_step] = {"image": image.detach(), "s2d": s2d.detach()}
        stale = self.global_step - self.options.interval - 1
        self.history.pop(stale, None)            # the reference never prunes (602 KB/frame host growth)

    def get_hist(self):
        h = self.history[self.global_step - self.options.interval]
        return h["image"], h["s2d"]

    # ------------------------------------------------------------------ teacher
    def update_teacher(self, teacher, model):
        ema_update(teacher.parameters(), model.parameters(), self.options.alpha)

    def cal_feature_diff(self, features_i, features_j):
        sims, mean = {}, 0
        for i, (a, b) in enumerate(zip(features_i, features_j)):
            c = F.cosine_similarity(a.flatten(), b.flatten(), dim=0, eps=1e-12)
            mean = mean + c
            sims[i] = {"cos": c}
        self.fit_losses["feat_sim/cos_sim"] = mean / i
        return sims

    # ------------------------------------------------------------------ losses
    def cal_shape_prior(self, pred_betas):
        return (pred_betas ** 2).sum(dim=-1).mean()

    def cal_pose_prior(self, pred_rotmat, betas=None):
        return pose_prior(pred_rotmat, self.gmm_f)

    def cal_s3d_loss(self, pred_s3d, gt_s3d, conf):
        gt = gt_s3d - ((gt_s3d[:, 2] + gt_s3d[:, 3]) / 2)[:, None, :]
        pr = pred_s3d - ((pred_s3d[:, 2] + pred_s3d[:, 3]) / 2)[:, None, :]
        return (conf * (pr - gt) ** 2).mean()

    def cal_teacher_loss(self, image, pred_rotmat, pred_shape, pred_s2d, pred_s3d):
        with torch.no_grad():
            t_rot, t_shape, t_cam = self.teacher(image)
            t_s3d = self.decode_smpl_params(t_rot, t_shape)["s3d"]
            t_s2d = self.projection(t_cam, t_s3d)["normed"]
        terms = dict(s2dloss=F.mse_loss(pred_s2d, t_s2d), s3dloss=F.mse_loss(t_s3d, pred_s3d),
                     shape_loss=F.mse_loss(pred_shape, t_shape), pose_loss=F.mse_loss(pred_rotmat, t_rot))
        loss = terms["s2dloss"] * 5 + terms["s3dloss"] * 5 + terms["shape_loss"] * 0.001 + terms["pose_loss"] * 1
        for k, v in terms.items():
            self.fit_losses[f"teacher/{k}"] = v
        self.fit_losses["teacher/loss"] = loss
        return loss

    def cal_motion_loss(self, model, pred_s2d, gt_s2d, prefix="ul"):
        hist_image, hist_s2d = self.get_hist()
        h_rot, h_shape, h_cam = model(hist_image)
        h_s3d = self.decode_smpl_params(h_rot, h_shape)["s3d"]
        h_s2d = self.projection(h_cam, h_s3d)["normed"]
        pred_motion = pred_s2d - h_s2d[:, 25:]
        gt_motion = gt_s2d[:, :, :-1] - hist_s2d[:, 25:, :-1]
        conf = ((hist_s2d[:, 25:, -1:] + gt_s2d[:, :, -1:]) == 2).float()
        loss = (((pred_motion - gt_motion) ** 2) * conf).mean()
        self.fit_losses[f"{prefix}/motion_loss"] = loss
        return loss

    def _teacher_term(self, image, rot, shape, cam, s3d):
        """cal_teacher_loss on the term kernel (same logging keys)."""
        with torch.no_grad():
            t_rot, t_shape, t_cam = self.teacher(image)
            t_s3d = self.decode_smpl_params(t_rot, t_shape)["s3d"]
        loss, comps = teacher_term(rot, shape, cam, s3d, t_rot, t_shape, t_cam, t_s3d)
        for i, k in enumerate(("s2dloss", "s3dloss", "shape_loss", "pose_loss")):
            self.fit_losses[f"teacher/{k}"] = comps[i]
        self.fit_losses["teacher/loss"] = loss
        return loss

    def _motion_term(self, model, rot, shape, cam, s3d, gt_keypoints_2d, prefix="ul"):
        """cal_motion_loss on the term kernel: the history frame through the same weights, both passes differentiable."""
        hist_image, hist_s2d = self.get_hist()
        h_rot, h_shape, h_cam = model(hist_image)
        h_s3d = self.decode_smpl_params(h_rot, h_shape)["s3d"]
        loss, _ = motion_term(rot, shape, cam, s3d, h_cam, h_s3d, gt_keypoints_2d, hist_s2d)
        self.fit_losses[f"{prefix}/motion_loss"] = loss
        return loss

    def adapt_on_labeled_data(self, model, batch, prefix="ll", kernels=False):
        from .geometry import batch_rodrigues
        gt_s2d = batch["keypoints"]
        conf = gt_s2d[:, 25:, -1:].clone()
        rot, shape, cam, feats = model(batch["img"], need_feature=True)
        s3d = self.decode_smpl_params(rot, shape)["s3d"]
        gt_rot = batch_rodrigues(batch["pose"].view(-1, 3)).view(-1, 24, 3, 3)
        assert batch["pose_3d"].shape[1] == 24
        if kernels:
            loss, comps = labelled_term(rot, shape, cam, s3d, gt_s2d, gt_rot, batch["betas"], batch["pose_3d"])
            terms = dict(labled_s2dloss=comps[0], labled_s3dloss=comps[1], labled_shape_loss=comps[2], labled_pose_loss=comps[3])
        else:
            s2d = self.projection(cam, s3d)["normed"]
            terms = dict(labled_s2dloss=(((s2d[:, 25:] - gt_s2d[:, 25:, :-1]) ** 2) * conf).mean(),
                         labled_s3dloss=self.cal_s3d_loss(s3d[:, 25:], batch["pose_3d"][:, :, :-1], conf),
                         labled_shape_loss=F.mse_loss(shape, batch["betas"]), labled_pose_loss=F.mse_loss(rot, gt_rot))
            loss = (terms["labled_s2dloss"] * 5 + terms["labled_s3dloss"] * 5 + terms["labled_shape_loss"] * 0.001
                    + terms["labled_pose_loss"] * 1)
        for k, v in terms.items():
            self.fit_losses[f"{prefix}/{k}"] = v
        self.fit_losses[f"{prefix}/labled_loss"] = loss
        return loss, feats

    def _level(self, level, image, gt_keypoints_2d, h36m_batch, learner):
        o = self.options
        tag = "ll" if level == "lower" else "ul"
        quiet = getattr(self, "_replay", None) is not None     # a second-order re-evaluation: no logging, same exemplars
        loss = None
        s2d = None
        log = {} if quiet else self.fit_losses
        fused = getattr(o, "fused_level", 1) and getattr(o, f"use_frame_losses_{level}")
        if fused:
            # model -> SMPL -> frame-loss head as ONE autograd node (fused_level.py); same numbers as the
            # three-module composition below, which stays for use_frame_losses_* = 0 and as the cross-check
            loss, comps, rot, shape, cam, s3d, _vts, feats = level_forward(
                learner, self.smpl_neutral, self.gmm_f, image, gt_keypoints_2d, o.s2dloss_weight, o.shape_prior_weight,
                o.pose_prior_weight)
        else:
            clear_last_forward()                 # (a remembered fused forward must not be mistaken for this level's: ADVICE r3)
            rot, shape, cam, feats = learner(image, need_feature=True)
            smpl_out = self.decode_smpl_params(rot, shape)
            s3d = smpl_out["s3d"]
        if not quiet:
            self._level_pred = (rot.detach(), shape.detach(), cam.detach())     # what inference() with these weights returns
        if getattr(o, f"use_frame_losses_{level}"):
            if not fused:
                loss, comps = frame_losses(rot, shape, cam, s3d, gt_keypoints_2d, self.gmm_f, o.s2dloss_weight,
                                           o.shape_prior_weight, o.pose_prior_weight)
            if not quiet:
                if level == "lower":
                    self.kp2dlosses_lower.append(comps[0])
                else:
                    self.kp2dlosses_upper[self.global_step] = comps[0]
            log[f"{tag}/s2dloss"], log[f"{tag}/shape_prior"] = comps[0], comps[1]
            log[f"{tag}/pose_prior"], log[f"{tag}/unlabelloss"] = comps[2], loss.detach()
        keep, self.fit_losses = self.fit_losses, log           # the term helpers below log into self.fit_losses
        try:
            # teacher / motion / labelled terms: one value + gradient launch each (losses._AuxTerms) - the same kernel the native
            # stepper issues; --term_kernels 0 (or a batch beyond its 16 samples) composes them from torch ops as the reference does
            kern = bool(getattr(o, "term_kernels", 1)) and image.shape[0] <= AUX_MAX_BATCH
            if getattr(o, f"use_temporal_losses_{level}"):
                if not kern:
                    s2d = self.projection(cam, s3d)["normed"]
                if o.use_meanteacher:
                    t = (self._teacher_term(image, rot, shape, cam, s3d) if kern else self.cal_teacher_loss(image, rot, shape, s2d, s3d))
                    t = t * o.teacherloss_weight
                    loss = t if loss is None else loss + t
                if o.use_motion and (self.global_step - o.interval) > 0:
                    m = (self._motion_term(learner, rot, shape, cam, s3d, gt_keypoints_2d, prefix="ul") if kern else
                         self.cal_motion_loss(learner, s2d[:, 25:], gt_keypoints_2d[:, 25:], prefix="ul"))
                    loss = loss + m * o.motionloss_weight
            if o.retrieval:
                h36m_batch = self._replay["h36m"] if quiet else self.retrieval(feats[5])
            self._last_h36m = h36m_batch
            if getattr(o, f"{level}_level_mixtrain"):
                lab, _ = self.adapt_on_labeled_data(learner, h36m_batch, prefix=tag,
                                                    kernels=kern and h36m_batch["img"].shape[0] <= AUX_MAX_BATCH)
                loss = loss + lab * o.labelloss_weight
        finally:
            self.fit_losses = keep
        # the reference builds the sum with in-place `loss += ...` on the tensor it stored as
        # '<tag>/unlabelloss' (base_adaptor.py:244,250,254,266), so what it logs under that key is this total
        log[f"{tag}/total"] = loss.detach()
        return loss, feats

    def level_closure(self, level, image, gt_keypoints_2d, h36m_batch):
        """The loss `_level` just returned, as a function of the learner (for MAML.adapt(..., closure=) in
        second-order mode): same data, same retrieved exemplars, nothing logged."""
        used = getattr(self, "_last_h36m", None)

        def closure(learner):
            self._replay = dict(h36m=used)
            try:
                return self._level(level, image, gt_keypoints_2d, h36m_batch, learner)[0]
            finally:
                self._replay = None
        return closure

    def level_hvp_factory(self, level, image, gt_keypoints_2d, learner):
        """With --hvp exact: theta -> (v -> H v) for the level `_level` just evaluated (dynaboa_amd/hvp.py): frame-loss levels
        through the kernels' own head gradient, levels with teacher / motion / labelled terms through the multi-pass form;
        None (MAML.adapt then differences the closure's gradient) with --hvp fd."""
        o = self.options
        if getattr(o, "hvp", "fd") != "exact":
            return None
        from .hvp import frame_level_hvp, general_level_hvp
        hmr = getattr(learner, "module", learner)
        other = getattr(o, f"use_temporal_losses_{level}") or getattr(o, f"{level}_level_mixtrain")
        if not other and getattr(o, f"use_frame_losses_{level}"):
            # frame losses only (the benchmarked second-order configuration): the head's gradient straight from the kernels
            return lambda theta: frame_level_hvp(hmr, self.smpl_neutral, self.gmm_f, theta, image, gt_keypoints_2d, o.s2dloss_weight,
                                                 o.shape_prior_weight, o.pose_prior_weight)
        if getattr(o, "hvp_terms", "all") != "all":
            return None                      # --hvp_terms frame: teacher / motion / labelled levels keep the difference quotient
        used = getattr(self, "_last_h36m", None)
        return lambda theta: general_level_hvp(self, level, hmr, theta, image, gt_keypoints_2d, used)

    def lower_level_adaptation(self, image, gt_keypoints_2d, h36m_batch, learner=None):
        return self._level("lower", image, gt_keypoints_2d, h36m_batch, learner)

    def upper_level_adaptation(self, image, gt_keypoints_2d, h36m_batch, learner=None):
        return self._level("upper", image, gt_keypoints_2d, h36m_batch, learner)

    # ------------------------------------------------------------------ stubs the drivers override
    def excute(self):
        pass

    def adaptation(self, batch):
        pass

    def inference(self, batch, model, need_feature=False):
        pass

    def write_summaries(self, losses):
        self.last_summaries = losses
