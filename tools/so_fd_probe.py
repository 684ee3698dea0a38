"""Second-order MAML, finite-difference step sweep: outer-gradient error of frame 0 against the reference's second-order
goldens (learn2learn first_order=False) for several MAML.fd_rel values, at inner_step 2 and 3."""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import golden
from dynaboa_amd import assets, benchmark as DB
from dynaboa_amd.base_adaptor import synthetic_bundle
from dynaboa_amd.maml import MAML
FRAME_ONLY = dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0, use_motion=0, dynamic_boa=0, use_temporal_losses_upper=0)
for K in (2, 3):
    gso, gfo = golden(f"g5_so_inner{K}_frameonly.npz"), golden(f"g5_fo_inner{K}_frameonly.npz")
    names = [str(x) for x in gso["names"]]
    gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
    for fd in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        MAML.fd_rel = fd
        o = DB.parser.parse_args([])
        for k, v in dict(FRAME_ONLY, inner_step=K, second_order=1).items(): setattr(o, k, v)
        ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=False, randomize_norm=True, smpl_seed=0), device="cuda:0")
        ad.reset_records(1); ad.global_step = 0; ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(0, 1, seed=22).items()}
        ad.model.eval(); ad.adaptation(batch)
        hmr = ad.model.module
        g1 = hmr._layout1.unpack(ad.optimizer.state[hmr.theta]["exp_avg"] / (1 - ad.options.beta1))
        gn = np.array([float(g1[k].double().norm()) for k in names])
        err = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
        sl = []
        for k in ("conv1.weight", "layer2.0.conv2.weight", "layer4.0.conv2.weight", "fc1.weight"):
            x = g1[k].flatten()[:256].double().cpu().numpy(); r = gso["g1_" + k]
            sl.append(float(np.abs(x - r).max() / np.abs(r).max()))
        print("K", K, "fd_rel", fd, "norm err median %.2e max %.2e (FO-SO gap median %.2e)" % (np.median(err), err.max(), np.median(gap)), "slice err", ["%.1e" % v for v in sl])
