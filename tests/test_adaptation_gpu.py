"""End-to-end parity of the drop-in Adaptor (dynaboa_amd.benchmark) on cuda:0 against goldens g5_*
= the REFERENCE's own Adaptor.adaptation() run frame after frame (tools/make_golden.py).
Because lr = 3e-6 moves outputs by ~1e-5 per frame, predictions alone would pass with a no-op
adapter: the Adam moments and the (theta_after - theta_before) deltas are compared too."""
import numpy as np
import pytest
import torch

from conftest import cosine, golden, rel_err

pytestmark = pytest.mark.gpu

FRAME_ONLY = dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0, use_motion=0,
                  dynamic_boa=0, use_temporal_losses_upper=0)
STREAMS = {
    "fo_inner3_frameonly": (dict(FRAME_ONLY, inner_step=3), False),
    "fo_inner1_frameonly_identity": (dict(FRAME_ONLY, inner_step=1), True),
    "fo_inner1_full": (dict(inner_step=1, interval=2, optim_steps=2), False),
    "fo_inner1_full_forced": (dict(inner_step=1, interval=2, optim_steps=2, cos_sim_threshold=-1.0), False),
}
SLICE_PARAMS = ["conv1.weight", "layer1.0.conv2.weight", "layer2.0.conv2.weight", "layer3.5.conv1.weight",
                "layer4.0.conv2.weight", "layer4.2.bn3.weight", "fc1.weight", "fc2.weight", "decpose.weight",
                "decpose.bias", "deccam.bias"]


def make_adaptor(opts_over, identity_pose, deferred=0):
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    o = DB.parser.parse_args([])
    for k, v in opts_over.items():
        setattr(o, k, v)
    o.deferred_metrics = deferred
    bundle = synthetic_bundle(seed=22, identity_pose=identity_pose, randomize_norm=True, smpl_seed=0)
    return DB.Adaptor(o, bundle, device="cuda:0"), bundle


@pytest.mark.parametrize("tag", list(STREAMS))
def test_stream_matches_reference(tag):
    from dynaboa_amd import assets
    g = golden(f"g5_{tag}.npz")
    opts, ident = STREAMS[tag]
    ad, bundle = make_adaptor(opts, ident)
    n = int(g["nframes"])
    ad.reset_records(n)
    hmr = ad.model.module
    theta0 = hmr.theta.detach().clone()
    for step in range(n):
        ad.global_step = step
        ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(step, 1, seed=22).items()}
        ad.model.eval()
        mpjpe, pampjpe, pve = ad.adaptation(batch)
        up = float(ad.fit_losses["ul/total"])      # golden 'ul/unlabelloss' aliases the in-place total (see base_adaptor._level)
        assert abs(up - g["upper_loss"][step]) < 1e-4 * abs(g["upper_loss"][step]), (step, up, g["upper_loss"][step])
        if opts.get("dynamic_boa", 1):
            assert ad.optim_step_record[-1] == int(g["extra_steps"][step])
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
            j = ad.decode_smpl_params(r, s)["s3d"]
        for k, v in dict(rotmat=r, shape=s, cam=c, joints=j).items():
            assert rel_err(v.cpu().numpy(), g[f"pred{step}_{k}"]) < 1e-3, (step, k)     # north_star: 1e-3 rel
        assert abs(float(np.mean(mpjpe)) - g["mpjpe"][step]) < 1e-3 * g["mpjpe"][step]
        assert abs(float(np.mean(pampjpe)) - g["pampjpe"][step]) < 2e-3 * g["pampjpe"][step]
        assert abs(float(pve) - g["pve"][step]) < 1e-3 * g["pve"][step]
    assert_final_state_matches_golden(ad, g, theta0, opts, tag=tag)


GATED = ["fo_inner1_full_gated", "fo_inner1_full_gated_b", "fo_inner1_full_gated_c"]


def assert_gate_matches_golden(ad, g, step):
    """Every check of the dynamic-BOA gate in frame `step` against the reference's (golden keys gate_*, tools/make_golden.py g5_gated):
    the number of checks, the step count the loop left with (dynaboa_benchmark.py:161-192), and 1 - cos(features[12]) of each check on
    the reference's side of the threshold with at least half the reference's distance from it to spare, the value within 1 %; all 15
    cosines against the reference's float64 evaluation (3e-6).  -> the largest deviation as a fraction of the check's distance from the
    threshold."""
    thr, margin = float(g["gate_threshold"]), float(g["gate_margin"])
    extra = int(g["extra_steps"][step])
    assert ad.optim_step_record[-1] == extra, (step, ad.optim_step_record[-1], extra)
    nchk = int(g["gate_checks"][step])
    sims = ad.feat_sims[step]
    assert len(sims) == nchk, (step, len(sims), nchk)
    worst = 0.0
    for k in range(nchk):
        ours = np.array([float(sims[k][i]["cos"]) for i in range(15)], np.float64)
        ref = g["gate_cos"][step, k]
        refd = float(g["gate_1mcos12"][step, k])
        dev = abs((1.0 - ours[12]) - refd)
        worst = max(worst, dev / abs(refd - thr))
        # same decision, with at least half of the reference's own distance from the threshold left (>= gate_margin x threshold for every
        # check of the run), and the value itself within 1 % (the weights these features come from carry up to 47 Adam steps of fp32
        # trajectory noise by the last frames: 0.3 % measured at frame 8)
        assert ((1.0 - ours[12]) > thr) == (refd > thr), (step, k)
        assert dev < 0.5 * abs(refd - thr), (step, k, 1.0 - ours[12], refd, thr)
        assert dev < 1e-2 * refd, (step, k, 1.0 - ours[12], refd)
        # the other 14 cosines: against the reference's values evaluated in float64 on its own features where the golden has them
        # (gate_cos64; F.cosine_similarity in fp32 carries up to ~2e-5 of summation error on the 8e5-element features - it returns
        # 1.000017 for feature 0), else against the fp32 values at that noise level
        if "gate_cos64" in g.files:
            np.testing.assert_allclose(ours, g["gate_cos64"][step, k], atol=3e-6, rtol=0)
        np.testing.assert_allclose(ours, ref, atol=5e-5, rtol=0)
    return worst


@pytest.mark.parametrize("native", [1, 0], ids=["native_stepper", "autograd_path"])
@pytest.mark.parametrize("tag", GATED)
def test_dynamic_boa_gate_leaves_by_convergence_as_the_reference_does(tag, native):
    """VERDICT r5 item 1.  The reference's Adaptor.adaptation() at its LITERAL defaults (inner_step 1, interval 5 - the motion term is
    live from frame 5 on -, optim_steps 7, every term on) over 10 frames, with a gate threshold chosen on the reference run so that the
    loop is never entered on some frames, LEAVES BY CONVERGENCE (1 - cos <= threshold after >= 1 extra step) on others and runs into the
    optim_steps cut-off on the rest; every decision of the reference run is at least gate_margin (2 - 5 %) of the threshold away from
    it, fp32 noise on 1 - cos being ~0.1 %.  Asserted per frame: the step count, every check's cosines, losses, predictions, metrics;
    at the end the Adam state (step count = 10 + the extra steps taken) and the teacher."""
    from dynaboa_amd import assets
    g = golden(f"g5_{tag}.npz")
    steps = [int(x) for x in g["extra_steps"]]
    assert any(0 < x <= 7 for x in steps) and 0 in steps, steps          # the golden really has a convergence exit and a closed gate
    opts = dict(inner_step=1, cos_sim_threshold=float(g["gate_threshold"]), native_step=native)
    ad, _ = make_adaptor(opts, False)
    assert ad.options.interval == 5 and ad.options.optim_steps == 7      # the literal defaults
    n = int(g["nframes"])
    ad.reset_records(n)
    theta0 = ad.model.module.theta.detach().clone()
    worst = 0.0
    for step in range(n):
        ad.global_step = step
        ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(step, 1, seed=22).items()}
        ad.model.eval()
        mpjpe, pampjpe, pve = ad.adaptation(batch)
        assert (ad._native is not None and ad._native.full) == bool(native)
        worst = max(worst, assert_gate_matches_golden(ad, g, step))
        up = float(ad.fit_losses["ul/total"])
        assert abs(up - g["upper_loss"][step]) < 1e-4 * abs(g["upper_loss"][step]), (step, up, g["upper_loss"][step])
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
            j = ad.decode_smpl_params(r, s)["s3d"]
        for k, v in dict(rotmat=r, shape=s, cam=c, joints=j).items():
            assert rel_err(v.cpu().numpy(), g[f"pred{step}_{k}"]) < 1e-3, (step, k)
        assert abs(float(np.mean(mpjpe)) - g["mpjpe"][step]) < 1e-3 * g["mpjpe"][step]
        assert abs(float(np.mean(pampjpe)) - g["pampjpe"][step]) < 2e-3 * g["pampjpe"][step]
    print("gate %s: worst |d(1 - cos12)| as a fraction of the check's distance from the threshold %.3f (margin of the reference run %.2e)" % (tag, worst, float(g["gate_margin"])))
    assert int(ad.optimizer.state[ad.model.module.theta]["step"]) == n + sum(min(x, 7) for x in steps) == int(g["adam_steps"])
    assert_final_state_matches_golden(ad, g, theta0, dict(inner_step=1), tag=tag)


def assert_final_state_matches_golden(ad, g, theta0, opts, slices=True, tag=None, factor=None):
    """Adam step count, per-tensor norms of the Adam moments and of (theta_after - theta_before), sampled slices, teacher drift:
    the end-of-stream half of the reference parity gate (shared with tests/test_headline_gpu.py, tests/test_replica_full_gpu.py).

    Bounds: with `tag` the bounds are DATA - tests/golden/g5_<tag>_noise.npz holds how far two fp32 evaluations of
    the stream (the reference itself, the oracle) sit from the fp64 evaluation, per tensor; a tensor's bound is NOISE_FACTOR (3) x the
    largest such deviation in its class (conftest.noise_bounds).  The measured floor is what the old blanket bounds asserted without
    evidence: ReLU-mask flips of near-zero activations and Adam's sign-like step move the stem / layer1 GroupNorm affines by up to
    ~1 % between two correct fp32 runs after a handful of frames, everything from layer2 up by 1e-4 ... 1e-3.  Without `tag` (streams
    that have no noise file) the round-5 blanket bounds apply: norms 2e-2 (early GroupNorm affines 4e-2), deltas 5e-2, cosines 0.99."""
    from conftest import noise_bounds, GOLDEN
    import os
    hmr = ad.model.module
    st = ad.optimizer.state[hmr.theta]
    assert st["step"] == int(g["adam_steps"])
    L = hmr._layout1
    delta = L.unpack((hmr.theta.detach().double() - theta0.double()).float())
    m, v = L.unpack(st["exp_avg"]), L.unpack(st["exp_avg_sq"])
    names = [str(x) for x in g["names"]]
    dn = np.array([float(delta[k].double().norm()) for k in names])
    mn = np.array([float(m[k].double().norm()) for k in names])
    vn = np.array([float(v[k].double().norm()) for k in names])
    have_teacher = "teacher_delta_norms" in g.files and opts.get("use_meanteacher", 1)
    if have_teacher:
        td = L.unpack((ad.teacher.theta.detach().double() - theta0.double()).float())
        tn = np.array([float(td[k].double().norm()) for k in names])
    nb = noise_bounds(tag, names, **({} if factor is None else dict(factor=factor))) if tag and os.path.exists(os.path.join(GOLDEN, f"g5_{tag}_noise.npz")) else None
    if nb is not None:
        report = []
        for q, x, ref in (("m", mn, g["m_norms"]), ("v", vn, g["v_norms"]), ("d", dn, g["delta_norms"])) + \
                         ((("t", tn, g["teacher_delta_norms"]),) if have_teacher and "t" in nb else ()):
            e = np.abs(np.asarray(x) - ref) / ref
            b = nb[q]["nd"]
            i = int(np.argmax(e / b))
            report.append("%s: worst %.2e of a %.2e bound (%s), median %.2e" % (q, e[i], b[i], names[i], float(np.median(e))))
            bad = [(names[j], float(e[j]), float(b[j])) for j in range(len(names)) if e[j] >= b[j]]
            assert not bad, (tag, q, "norm deviation from the golden beyond 3 x the measured fp32 floor of the tensor's class", bad[:8])
        print("end-of-stream state vs golden %s (bounds = 3 x fp32-vs-fp64 floor): %s" % (tag, "; ".join(report)))
        for k in SLICE_PARAMS if slices else ():
            j = names.index(k)
            cm, cd = cosine(m[k].flatten()[:256], g["m_" + k]), cosine(delta[k].flatten()[:256], g["d_" + k])
            assert cm > nb["m"]["cos"][j], (tag, "m slice", k, cm, float(nb["m"]["cos"][j]))
            assert cd > nb["d"]["cos"][j], (tag, "delta slice", k, cd, float(nb["d"]["cos"][j]))
        return

    # norms: 2 % (ReLU-mask flips of near-zero activations perturb early-layer gradients at the 1e-3 level;
    # theta deltas are additionally quantised by fp32 rounding of p - 1e-5)
    def close(x, ref, tol, what):
        """per-tensor norms within `tol` for every tensor except the stem / layer1 GroupNorm affine tensors (4-64 floats each,
        fed by the ReLU-flip noise of the whole network above them over up to 12 Adam steps: measured 2.1 % on v), which get
        2 * tol.  The offenders are named in the failure message."""
        e = np.abs(np.asarray(x) - ref) / ref
        early = np.array([k.startswith("bn1.") or (k.startswith("layer1.") and (".bn" in k or "downsample.1" in k)) for k in names])
        bad = [(names[i], float(e[i])) for i in range(len(names)) if e[i] >= (2 * tol if early[i] else tol)]
        assert not bad, (what, bad[:8])
    close(mn, g["m_norms"], 2e-2, "m")
    close(vn, g["v_norms"], 2e-2, "v")
    close(dn, g["delta_norms"], 5e-2, "delta")
    for k in SLICE_PARAMS if slices else ():
        assert cosine(m[k].flatten()[:256], g["m_" + k]) > 0.99, k     # early-layer slices carry ReLU-flip noise
        assert cosine(delta[k].flatten()[:256], g["d_" + k]) > 0.99, k
    if have_teacher:
        np.testing.assert_allclose(tn, g["teacher_delta_norms"], rtol=5e-2)


def assert_first_frame_outer_gradient(ad, g, cos_min=0.9999, norm_tol=1e-3):
    """VERDICT r3 / SURVEY 8c: after the FIRST Adam step m = (1 - beta1) * g, so the outer gradient of frame 0 is exp_avg / (1 - beta1):
    per-tensor norms against the reference's `g1_norms` (median within norm_tol; every tensor within 10 x - the stem / layer1
    GroupNorm affines collect the ReLU-flip noise of the whole network) and the sampled slices `g1_<name>` at cosine >= cos_min."""
    hmr = ad.model.module
    st = ad.optimizer.state[hmr.theta]
    assert int(st["step"]) == 1
    g1 = hmr._layout1.unpack(st["exp_avg"] / (1 - ad.options.beta1))
    names = [str(x) for x in g["names"]]
    gn = np.array([float(g1[k].double().norm()) for k in names])
    err = np.abs(gn - g["g1_norms"]) / g["g1_norms"]
    sl = {k[3:]: cosine(g1[k[3:]].flatten()[:256].double().cpu().numpy(), g[k]) for k in g.files if k.startswith("g1_") and k != "g1_norms"}
    print("first-frame outer gradient: norm error median %.2e max %.2e; worst slice cosine %.6f" % (np.median(err), err.max(), min(sl.values())))
    assert np.median(err) < norm_tol and err.max() < 10 * norm_tol, (float(np.median(err)), [(names[i], float(err[i])) for i in np.argsort(-err)[:5]])
    bad = {k: v for k, v in sl.items() if v < cos_min}
    assert not bad, bad


@pytest.mark.parametrize("native", [1, 0], ids=["native_stepper", "autograd_path"])
def test_first_frame_first_order_outer_gradient_matches_reference(native):
    """Frame 0 of golden g5_fo_inner3_frameonly (the reference's Adaptor.adaptation, dynaboa_benchmark.py:126-157, first_order=True):
    the outer gradient itself - not only the Adam state after 4 frames - at cosine 0.9999 / norms 1e-3."""
    from dynaboa_amd import assets
    g = golden("g5_fo_inner3_frameonly.npz")
    opts, ident = STREAMS["fo_inner3_frameonly"]
    ad, _ = make_adaptor(dict(opts, native_step=native), ident)
    ad.reset_records(1)
    ad.global_step = 0
    ad.fit_losses = {}
    ad.model.eval()
    ad.adaptation({k: v.to(ad.device) for k, v in assets.make_frame(0, 1, seed=22).items()})
    assert (ad._native is not None) == bool(native)
    assert_first_frame_outer_gradient(ad, g)


@pytest.mark.parametrize("native", [1, 0], ids=["native_stepper", "autograd_path"])
def test_first_frame_outer_gradient_of_the_default_term_set_matches_reference(native):
    """The same for the reference's DEFAULT flags (golden g5_fo_inner1_full: teacher term + labelled exemplars in both levels, the
    dynamic-BOA gate - closed on frame 0 in the reference run too, so frame 0 ends after ONE Adam step): the first outer gradient
    of the full term set, tensor by tensor, against the reference's own (base_adaptor.py:222-398, dynaboa_benchmark.py:126-193)."""
    from dynaboa_amd import assets
    g = golden("g5_fo_inner1_full.npz")
    assert int(g["extra_steps"][0]) == 0
    opts, ident = STREAMS["fo_inner1_full"]
    ad, _ = make_adaptor(dict(opts, native_step=native), ident)
    ad.reset_records(1)
    ad.global_step = 0
    ad.fit_losses = {}
    ad.model.eval()
    ad.adaptation({k: v.to(ad.device) for k, v in assets.make_frame(0, 1, seed=22).items()})
    assert (ad._native is not None and ad._native.full) == bool(native)
    assert ad.optim_step_record[-1] == 0
    assert_first_frame_outer_gradient(ad, g)


def test_deferred_metrics_equal_immediate():
    from dynaboa_amd import assets
    opts, ident = STREAMS["fo_inner1_frameonly_identity"]
    frames = [assets.make_frame(s, 1, seed=22) for s in range(2)]
    ad1, _ = make_adaptor(opts, ident, deferred=0)
    r1 = ad1.excute(frames, nframes=2)
    ad2, _ = make_adaptor(opts, ident, deferred=1)
    r2 = ad2.excute(frames, nframes=2)
    np.testing.assert_allclose(np.concatenate(r1["pampjpe"]), np.concatenate([np.atleast_1d(x) for x in r2["pampjpe"]]), rtol=1e-4)
    np.testing.assert_allclose(np.concatenate(r1["mpjpe"]), np.concatenate([np.atleast_1d(x) for x in r2["mpjpe"]]), rtol=1e-5)


def test_side_stream_overlap_changes_nothing():
    """Metric / feature forwards on a side HIP stream: same kernels, same inputs -> identical state
    and metrics (guards the event / allocator plumbing)."""
    from dynaboa_amd import assets
    frames = [assets.make_frame(s, 1, seed=22) for s in range(3)]
    outs = []
    for overlap in (0, 1, 2):
        opts, ident = STREAMS["fo_inner3_frameonly"]
        ad, _ = make_adaptor(dict(opts, overlap_metrics=overlap), ident, deferred=1)
        r = ad.excute(frames, nframes=3)
        outs.append((ad.model.module.theta.detach().clone(), r))
    for other in outs[1:]:
        assert torch.equal(outs[0][0], other[0])
        for k in ("mpjpe", "pampjpe", "pve"):
            np.testing.assert_array_equal(np.array(outs[0][1][k], dtype=np.float64).ravel(), np.array(other[1][k], dtype=np.float64).ravel())
    full, ident = STREAMS["fo_inner1_full_forced"]
    outs = []
    for overlap in (0, 1, 2):
        ad, _ = make_adaptor(dict(full, overlap_metrics=overlap), ident, deferred=1)
        ad.excute(frames, nframes=3)
        outs.append(ad.model.module.theta.detach().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_engine_graph_cache_is_bit_identical():
    """Cached-hipGraph replays of whole forward/backward calls must reproduce the eager path exactly."""
    from dynaboa_amd import _lib
    from dynaboa_amd import assets
    from dynaboa_amd.hmr import get_layout
    L = get_layout(1)
    lib = _lib.load()
    frames = [assets.make_frame(s, 1, seed=22) for s in range(14)]
    outs = []
    for on in (0, 1):
        lib.dyb_hmr_set_graph_mode(L.plan, on)
        opts, ident = STREAMS["fo_inner3_frameonly"]
        ad, _ = make_adaptor(opts, ident, deferred=1)
        before = L.graph_stats()
        r = ad.excute(frames, nframes=14)
        after = L.graph_stats()
        outs.append((ad.model.module.theta.detach().clone(), np.array(r["pampjpe"], dtype=np.float64).ravel(),
                     after["replays"] - before["replays"]))
    lib.dyb_hmr_set_graph_mode(L.plan, 1 if L.graphs else 0)
    assert outs[0][2] == 0                     # mode off: never replays
    print("graph replays with mode on:", outs[1][2], L.graph_stats())   # address recurrence is allocator-dependent
    assert torch.equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_state_dict_roundtrip_and_missing_extension_is_loud(tmp_path):
    from dynaboa_amd import _lib, assets
    from dynaboa_amd.hmr import hmr
    from dynaboa_amd.maml import MAML
    ck = assets.make_synthetic_checkpoint(5, randomize_norm=True)
    model = MAML(hmr(assets.make_smpl_mean_params(), seed=1), lr=8e-6).to("cuda:0")
    model.load_state_dict(ck["model"], strict=True)
    sd = model.state_dict()
    assert set(sd) == set(ck["model"])
    for k, v in ck["model"].items():
        assert torch.equal(sd[k].cpu().reshape(v.shape), v), k
    with pytest.raises(RuntimeError):
        model.load_state_dict({"conv1.weight": torch.zeros(1)}, strict=True)
    # no fallback: with the library path broken, the loader raises instead of computing elsewhere
    saved, _lib._lib, _lib.LIB_PATH = _lib._lib, None, str(tmp_path / "nope.so")
    try:
        with pytest.raises(_lib.MissingExtension):
            _lib.load()
    finally:
        _lib._lib, _lib.LIB_PATH = saved, _lib.LIB_PATH.replace(str(tmp_path / "nope.so"), "")
        _lib.LIB_PATH = __import__("os").path.join(__import__("os").path.dirname(_lib.__file__), "libdynaboa_hip.so")


def test_hmr_forward_explicit_init_and_n_iter_vs_reference(ckpt_rand):
    """HMR.forward(x, init_pose, init_shape, init_cam, n_iter=2) (reference model/hmr.py:127,132-137) against
    golden g3 'alt_*' produced by the reference module itself."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr import hmr
    g = golden("g3_hmr.npz")
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    m = hmr(mp, seed=1).to("cuda:0").eval()
    m.load_state_dict(ckpt_rand, strict=True)
    img = assets.make_frame(0, batch_size=2, seed=22)["image"][:1].to("cuda:0")
    with torch.no_grad():
        r, s, c = m(img, init_pose=m.init_pose * 0.9, init_shape=m.init_shape + 0.1, init_cam=m.init_cam * 1.1, n_iter=2)
        r3, s3, c3, feats = m(img, need_feature=True)
    assert rel_err(r.cpu().numpy(), g["alt_rotmat"]) < 1e-5
    assert rel_err(s.cpu().numpy(), g["alt_shape"]) < 1e-5 and rel_err(c.cpu().numpy(), g["alt_cam"]) < 1e-5
    assert len(feats) == 15 and list(feats[1].shape) == [1, 256, 56, 56] and list(feats[12].shape) == [1, 1024]
    assert rel_err(r3.cpu().numpy(), g["rotmat"][:1]) < 1e-5
    # train(): live Dropout after fc1 / fc2 (model/hmr.py:165,169) - fresh masks per call, finite, different from eval
    m.train()
    with torch.no_grad():
        ra, sa, ca = m(img)
        rb, sb, cb = m(img)
    assert torch.isfinite(ra).all() and not torch.equal(sa, sb) and not torch.equal(sa, s3)
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(img)[1], s3)


def test_batch4_frame_matches_oracle(gmm_t, smpl_tabs):
    """One full bilevel step at batch 4 (SURVEY 8d config 3 size class) against the CPU oracle:
    predictions after the step, and the Adam first moment (= 0.5 * outer gradient after step 1)."""
    from dynaboa_amd import assets
    from oracle import ref_cpu as O
    opts = dict(FRAME_ONLY, inner_step=2, batch_size=4)
    ad, bundle = make_adaptor(opts, False)
    ad.reset_records(1)
    ad.global_step = 0
    frame = assets.make_frame(3, 4, seed=22)
    ad.model.eval()
    ad.adaptation({k: v.to(ad.device) for k, v in frame.items()})
    with torch.no_grad():
        r, s, c = ad.model(frame["image"].to(ad.device))
        j = ad.decode_smpl_params(r, s)["s3d"]
    sd = {k.replace("module.", ""): v for k, v in bundle.checkpoint["model"].items()}
    ref = O.Adapter(sd, O.smpl_tables_to_torch(smpl_tabs), gmm_t, opts)
    rec = ref.adapt_frame(frame)
    for name, a, b in (("rotmat", r, rec["pred"]["rotmat"]), ("shape", s, rec["pred"]["shape"]), ("cam", c, rec["pred"]["cam"]),
                       ("joints", j, rec["pred"]["joints"])):
        assert rel_err(a.cpu().numpy(), b.numpy()) < 1e-3, name
    hmr_m = ad.model.module
    m1 = hmr_m._layout1.unpack(ad.optimizer.state[hmr_m.theta]["exp_avg"])
    for k in ("conv1.weight", "layer2.0.conv2.weight", "layer4.2.conv3.weight", "fc1.weight", "decpose.bias"):
        a, b = m1[k].double().flatten() * 2, rec["outer_grad"][k].double().flatten()
        assert cosine(a, b) > 0.999, k
        assert abs(float(a.norm() / b.norm()) - 1) < 2e-2, k


def test_second_order_matches_reference_second_order():
    """second_order=1 (finite-difference Hessian-vector products over the first-order engine, dynaboa_amd/maml.py)
    against the reference run with learn2learn first_order=False (golden g5_so_inner2_frameonly): losses, predictions,
    and the outer gradient of the first frame - which must be much closer to the second-order golden than the
    first-order golden of the same frame is (they differ by 16-40 % per tensor)."""
    from dynaboa_amd import assets
    gso, gfo = golden("g5_so_inner2_frameonly.npz"), golden("g5_fo_inner2_frameonly.npz")
    ad, bundle = make_adaptor(dict(FRAME_ONLY, inner_step=2, second_order=1, hvp="fd"), False)
    n = int(gso["nframes"])
    ad.reset_records(n)
    hmr = ad.model.module
    L = hmr._layout1
    names = [str(x) for x in gso["names"]]
    for step in range(n):
        ad.global_step = step
        ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(step, 1, seed=22).items()}
        ad.model.eval()
        ad.adaptation(batch)
        up = float(ad.fit_losses["ul/total"])
        assert abs(up - gso["upper_loss"][step]) < 1e-4 * abs(gso["upper_loss"][step]), (step, up)
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
        for k, v in dict(rotmat=r, shape=s, cam=c).items():
            assert rel_err(v.cpu().numpy(), gso[f"pred{step}_{k}"]) < 1e-3, (step, k)
        if step == 0:
            st = ad.optimizer.state[hmr.theta]
            g1 = L.unpack(st["exp_avg"] / (1 - ad.options.beta1))
            gn = np.array([float(g1[k].double().norm()) for k in names])
            err_so = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
            gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
            print("second-order grad-norm error: median %.2e max %.2e; FO-vs-SO gap: median %.2e" % (np.median(err_so), err_so.max(), np.median(gap)))
            assert np.median(err_so) < 3e-3 and err_so.max() < 2e-2          # measured 3.7e-4..1.2e-3 / 5e-3..8e-3 (profiles/r02_so_fd_sweep.txt)
            assert np.median(err_so) < 0.1 * np.median(gap)
            for k in SLICE_PARAMS:
                x = g1[k].flatten()[:256].double().cpu().numpy()
                assert rel_err(x, gso["g1_" + k]) < 0.15 * rel_err(gfo["g1_" + k], gso["g1_" + k]) + 2e-2, k


def test_second_order_needs_closure():
    from dynaboa_amd import assets
    ad, _ = make_adaptor(dict(FRAME_ONLY, inner_step=1, second_order=1), False)
    batch = {k: v.to(ad.device) for k, v in assets.make_frame(0, 1, seed=22).items()}
    ad.model.eval()
    learner = ad.model.clone()
    loss, _ = ad.lower_level_adaptation(batch["image"], batch["smpl_j2d"], None, learner)
    with pytest.raises(NotImplementedError, match="closure"):
        learner.adapt(loss)


def test_fused_level_node_matches_three_module_composition():
    """fused_level=1 (one autograd node per level: HMR -> SMPL -> frame-loss head) against fused_level=0 (HMR.forward,
    SMPL.forward, losses.frame_losses composed through autograd): frame-only levels must agree bit for bit (same
    kernels, and with an incoming gradient of 1 the fused gradient assembly is the same two-term sum); the full loss
    set, whose teacher / motion / label gradients enter the node as external gradients, to rounding."""
    from dynaboa_amd import assets
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(3)]
    res = {}
    for name, (opts, ident) in dict(frame=STREAMS["fo_inner3_frameonly"], full=STREAMS["fo_inner1_full"]).items():
        for fused in (0, 1):
            ad, _ = make_adaptor(dict(opts, fused_level=fused), ident)
            ad.excute(frames, nframes=3)
            res[(name, fused)] = (ad.model.module.theta.detach().clone(),
                                  ad.optimizer.state[ad.model.module.theta]["exp_avg"].clone())
    assert torch.equal(res[("frame", 0)][0], res[("frame", 1)][0])
    assert torch.equal(res[("frame", 0)][1], res[("frame", 1)][1])
    m0, m1 = res[("full", 0)][1], res[("full", 1)][1]
    assert float((m0 - m1).norm() / m0.norm()) < 2e-3          # ReLU-flip noise class, see DESIGN.md 4
    d0 = (res[("full", 0)][0] - res[("full", 1)][0]).abs()
    # three Adam steps of lr 3e-6: an element whose gradient is rounding noise moves by +-lr whatever its size, so two paths that differ
    # in summation order can part by up to 2 lr per step on such elements (6.9e-6 measured) - and ONLY on such elements: where
    # the two paths' first moments agree in sign and to a quarter of their size the weights must agree to the old, tight bound
    noise = (torch.sign(m0) != torch.sign(m1)) | ((m0 - m1).abs() > 0.25 * m0.abs())
    assert float(noise.float().mean()) < 0.02, float(noise.float().mean())
    assert float(d0[~noise].max()) < 5e-6, float(d0[~noise].max())
    assert float(d0.max()) < 1.9e-5


def test_shared_forwards_are_bit_identical():
    """share_forwards=1 takes the un-adapted feature forward and every per-inner-step inference() from the level
    forward with the same weights instead of recomputing them: weights, Adam state and every metric must be
    identical to the 9-forward schedule, for the frame-only stream and for the full loss set with the dynamic loop."""
    from dynaboa_amd import assets
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(3)]
    for opts, ident in (STREAMS["fo_inner3_frameonly"], STREAMS["fo_inner1_full_forced"]):
        outs = []
        for share in (0, 1):
            # native_step=0: this is about the autograd composition's own forward sharing (the native stepper always shares
            # and has its own identity test; its metric reductions are not bit-equal to torch's)
            ad, _ = make_adaptor(dict(opts, share_forwards=share, eval_lower=1, native_step=0), ident, deferred=1)
            res = ad.excute(frames, nframes=3)
            recs = sorted(((r["step"], r["tag"], float(np.ravel(r["mpjpe"])[0]), float(np.ravel(r["pampjpe"])[0]), r["pve"])
                           for r in ad.metric_records), key=lambda t: (t[0], str(t[1])))
            outs.append((ad.model.module.theta.detach().clone(), res, recs, ad.optim_step_record))
        assert torch.equal(outs[0][0], outs[1][0])
        for k in ("mpjpe", "pampjpe", "pve"):
            np.testing.assert_array_equal(np.array(outs[0][1][k], dtype=np.float64).ravel(), np.array(outs[1][1][k], dtype=np.float64).ravel())
        assert outs[0][3] == outs[1][3]
        assert len(outs[0][2]) == len(outs[1][2]) and len(outs[0][2]) >= 3 * (1 + opts["inner_step"])
        assert outs[0][2] == outs[1][2]                      # every inference() record, per-inner-step ones included


def test_second_order_full_loss_set_vs_oracle(gmm_t, smpl_tabs):
    """Second order with the reference's full default term set (teacher + labelled exemplars; the motion term needs
    history and is inactive on frame 0), i.e. the closure path with retrieved exemplars and external gradients on the
    level node: outer gradient of one frame against the CPU oracle in create_graph=True mode, and against the
    oracle's first-order gradient to show which of the two it follows."""
    from dynaboa_amd import assets
    from oracle import ref_cpu as O
    opts = dict(inner_step=1, interval=2, optim_steps=2, dynamic_boa=0)
    frame = assets.make_frame(0, 1, seed=22)
    grads = {}
    sd = None
    for so in (1, 0):
        if so:
            ad, bundle = make_adaptor(dict(opts, second_order=1), False)
            ad.reset_records(1)
            ad.global_step = 0
            ad.model.eval()
            ad.adaptation({k: v.to(ad.device) for k, v in frame.items()})
            hmr_m = ad.model.module
            ours = hmr_m._layout1.unpack(ad.optimizer.state[hmr_m.theta]["exp_avg"] / (1 - ad.options.beta1))
            sd = {k.replace("module.", ""): v for k, v in bundle.checkpoint["model"].items()}
        ref = O.Adapter(sd, O.smpl_tables_to_torch(smpl_tabs), gmm_t, opts, first_order=not so)
        ref.exemplar_fn = lambda step: assets.make_exemplars(step, ref.o["sample_num"])
        torch.set_num_threads(32)
        grads[so] = ref.adapt_frame(frame)["outer_grad"]
    names = ["conv1.weight", "layer2.0.conv2.weight", "layer3.5.conv1.weight", "layer4.2.conv3.weight", "fc1.weight", "decpose.weight"]
    e_so = np.array([rel_err(ours[k].cpu().numpy(), grads[1][k].numpy()) for k in names])
    gap = np.array([rel_err(grads[0][k].numpy(), grads[1][k].numpy()) for k in names])
    print("SO full-loss: err vs oracle-SO", e_so, " FO-vs-SO gap", gap)
    # measured: 4e-2 / 2e-2 / 6e-3 on conv1 / layer2 / layer3 (the difference quotient crosses ReLU kinks on the way
    # down 50 layers), 1e-5 from layer4 up - against first-vs-second-order gaps of 14-30 %
    assert (e_so < 6e-2).all(), e_so
    assert (e_so < 0.3 * gap + 5e-3).all(), (e_so, gap)


@pytest.fixture
def restore_switches():
    """Library switches changed by a test (dyb_set_option) go back to their defaults afterwards."""
    from dynaboa_amd import _lib
    yield
    lib = _lib.load()
    for name, dflt in ((b"k4", 1), (b"k4_bwd", 1), (b"k4_batch", 1), (b"rep_split", 0), (b"bf16", 0)):
        lib.dyb_set_option(name, dflt)


@pytest.mark.parametrize("k4_bwd", [1, 0])
def test_k4_backward_path_matches_reference_stream(k4_bwd, restore_switches):
    """k4_bwd=1 (default: data gradients of the small 1x1 layers carry the producer's GroupNorm-backward reduce in their
    epilogue) and k4_bwd=0 (two launches) on the 3-inner-step golden stream: same losses / predictions / Adam moments
    as the reference within the usual tolerances."""
    from dynaboa_amd import _lib, assets
    _lib.load().dyb_set_option(b"k4_bwd", k4_bwd)
    g = golden("g5_fo_inner3_frameonly.npz")
    opts, ident = STREAMS["fo_inner3_frameonly"]
    ad, _ = make_adaptor(opts, ident)
    n = int(g["nframes"])
    ad.reset_records(n)
    hmr = ad.model.module
    for step in range(n):
        ad.global_step = step
        ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(step, 1, seed=22).items()}
        ad.model.eval()
        ad.adaptation(batch)
        up = float(ad.fit_losses["ul/total"])
        assert abs(up - g["upper_loss"][step]) < 1e-4 * abs(g["upper_loss"][step]), (step, up)
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
        for k, v in dict(rotmat=r, shape=s, cam=c).items():
            assert rel_err(v.cpu().numpy(), g[f"pred{step}_{k}"]) < 1e-3, (step, k)
    st = ad.optimizer.state[hmr.theta]
    L = hmr._layout1
    m, v = L.unpack(st["exp_avg"]), L.unpack(st["exp_avg_sq"])
    names = [str(x) for x in g["names"]]
    np.testing.assert_allclose(np.array([float(m[k].double().norm()) for k in names]), g["m_norms"], rtol=2e-2)
    np.testing.assert_allclose(np.array([float(v[k].double().norm()) for k in names]), g["v_norms"], rtol=2e-2)


def test_batch8_mixtrain_matches_oracle(gmm_t, smpl_tabs):
    """BASELINE configs[2]: batch 8, lower- and upper-level labelled exemplars mixed in (S = 8 synthetic exemplars per
    level, reference base_adaptor.py:346-376 through dynaboa_benchmark.py:138-151) - one full bilevel step (2 inner +
    1 outer) against the CPU oracle: predictions after the step and the outer gradient.  Runs with the default dispatch
    (single-launch 1x1 kernels at batch > 1, k4_batch=1)."""
    from dynaboa_amd import assets
    from oracle import ref_cpu as O
    opts = dict(inner_step=2, batch_size=8, sample_num=8, retrieval=1, lower_level_mixtrain=1, upper_level_mixtrain=1,
                use_meanteacher=0, use_motion=0, dynamic_boa=0, use_temporal_losses_upper=0)
    ad, bundle = make_adaptor(opts, False)
    ad.reset_records(1)
    ad.global_step = 0
    frame = assets.make_frame(5, 8, seed=22)
    ad.model.eval()
    ad.adaptation({k: v.to(ad.device) for k, v in frame.items()})
    with torch.no_grad():
        r, s, c = ad.model(frame["image"].to(ad.device))
        j = ad.decode_smpl_params(r, s)["s3d"]
    sd = {k.replace("module.", ""): v for k, v in bundle.checkpoint["model"].items()}
    ref = O.Adapter(sd, O.smpl_tables_to_torch(smpl_tabs), gmm_t, opts)
    ref.exemplar_fn = lambda step: assets.make_exemplars(step, 8)
    torch.set_num_threads(max(1, min(64, (__import__("os").cpu_count() or 8) // 2)))
    rec = ref.adapt_frame(frame)
    assert abs(float(ad.fit_losses["ul/total"]) - rec["upper_loss"][0]) < 1e-4 * abs(rec["upper_loss"][0])
    for name, a, b in (("rotmat", r, rec["pred"]["rotmat"]), ("shape", s, rec["pred"]["shape"]), ("cam", c, rec["pred"]["cam"]),
                       ("joints", j, rec["pred"]["joints"])):
        assert rel_err(a.cpu().numpy(), b.numpy()) < 1e-3, name
    hmr_m = ad.model.module
    m1 = hmr_m._layout1.unpack(ad.optimizer.state[hmr_m.theta]["exp_avg"])
    for k in ("conv1.weight", "layer1.0.conv2.weight", "layer2.0.conv2.weight", "layer3.5.conv1.weight", "layer4.2.conv3.weight",
              "fc1.weight", "decpose.bias"):
        a, b = m1[k].double().flatten() * 2, rec["outer_grad"][k].double().flatten()
        assert cosine(a, b) > 0.999, k
        assert abs(float(a.norm() / b.norm()) - 1) < 2e-2, k


def test_second_order_inner3_matches_reference_second_order():
    """Second order with the difference-quotient Hessian-vector products (--hvp fd) at the BENCHMARKED depth (inner_step=3,
    BASELINE configs[1]; the exact form is test_second_order_inner3_exact_hvp_matches_reference_second_order) against the reference run with
    learn2learn first_order=False (golden g5_so_inner3_frameonly); the first-order golden of the same stream
    (g5_fo_inner3_frameonly) shows which of the two gradients the implementation follows."""
    from dynaboa_amd import assets
    gso, gfo = golden("g5_so_inner3_frameonly.npz"), golden("g5_fo_inner3_frameonly.npz")
    ad, bundle = make_adaptor(dict(FRAME_ONLY, inner_step=3, second_order=1, hvp="fd"), False)
    n = int(gso["nframes"])
    ad.reset_records(n)
    hmr = ad.model.module
    L = hmr._layout1
    names = [str(x) for x in gso["names"]]
    for step in range(n):
        ad.global_step = step
        ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(step, 1, seed=22).items()}
        ad.model.eval()
        ad.adaptation(batch)
        up = float(ad.fit_losses["ul/total"])
        assert abs(up - gso["upper_loss"][step]) < 1e-4 * abs(gso["upper_loss"][step]), (step, up)
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
        for k, v in dict(rotmat=r, shape=s, cam=c).items():
            assert rel_err(v.cpu().numpy(), gso[f"pred{step}_{k}"]) < 1e-3, (step, k)
        if step == 0:
            st = ad.optimizer.state[hmr.theta]
            g1 = L.unpack(st["exp_avg"] / (1 - ad.options.beta1))
            gn = np.array([float(g1[k].double().norm()) for k in names])
            err_so = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
            gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
            print("SO inner3 grad-norm error: median %.2e max %.2e; FO-vs-SO gap: median %.2e" % (np.median(err_so), err_so.max(), np.median(gap)))
            assert np.median(err_so) < 3e-3 and err_so.max() < 2e-2          # measured 3.7e-4..1.2e-3 / 5e-3..8e-3 (profiles/r02_so_fd_sweep.txt)
            assert np.median(err_so) < 0.1 * np.median(gap)


def test_second_order_inner3_exact_hvp_matches_reference_second_order():
    """--hvp exact (tangent passes through the network instead of a difference quotient of the whole level) at the benchmarked
    depth against the reference run with learn2learn first_order=False: per-tensor outer-gradient norms within 1e-3-class
    tolerances and cosine > 0.9999 on every sampled slice (emulator: median 1.8e-5, max 1.0e-3, min cosine 0.999996)."""
    from dynaboa_amd import assets
    from conftest import cosine
    gso, gfo = golden("g5_so_inner3_frameonly.npz"), golden("g5_fo_inner3_frameonly.npz")
    ad, bundle = make_adaptor(dict(FRAME_ONLY, inner_step=3, second_order=1, hvp="exact"), False)
    ad.reset_records(1)
    hmr = ad.model.module
    ad.global_step = 0
    ad.fit_losses = {}
    batch = {k: v.to(ad.device) for k, v in assets.make_frame(0, 1, seed=22).items()}
    ad.model.eval()
    ad.adaptation(batch)
    up = float(ad.fit_losses["ul/total"])
    assert abs(up - gso["upper_loss"][0]) < 1e-4 * abs(gso["upper_loss"][0])
    st = ad.optimizer.state[hmr.theta]
    g1 = hmr._layout1.unpack((st["exp_avg"] / (1 - ad.options.beta1)).cpu())
    names = [str(x) for x in gso["names"]]
    gn = np.array([float(g1[k].double().norm()) for k in names])
    err = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
    gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
    sl = {k[3:]: cosine(g1[k[3:]].flatten()[:256].numpy(), gso[k]) for k in gso.files if k.startswith("g1_") and k != "g1_norms"}
    print("exact SO inner3: grad-norm error median %.2e max %.2e (FO-SO gap median %.2e), min slice cosine %.6f" % (
        np.median(err), err.max(), np.median(gap), min(sl.values())))
    assert np.median(err) < 5e-4 and err.max() < 3e-3
    assert min(sl.values()) > 0.9999, sl


@pytest.mark.parametrize("hvp_terms", ["all", "frame"])
def test_second_order_full_loss_set_matches_reference_second_order(hvp_terms):
    """Second order on the reference's DEFAULT term set (labelled exemplars in the lower level; teacher + motion + exemplars in the
    upper; motion from frame 3 on) against the reference itself run with learn2learn first_order=False (golden g5_so_inner1_full,
    4 frames): upper losses, predictions, the first frame's outer gradient per tensor, final Adam state.  hvp_terms=all: exact
    Hessian-vector products for every level (the multi-pass form, dynaboa_amd/hvp.py general_level_hvp); hvp_terms=frame: levels
    with teacher / motion / labelled terms take the difference quotient (looser gradient bounds).  The first-order golden of the
    same stream (g5_fo_inner1_full) shows which gradient is being followed."""
    from dynaboa_amd import assets
    gso, gfo = golden("g5_so_inner1_full.npz"), golden("g5_fo_inner1_full.npz")
    opts = dict(inner_step=1, interval=2, optim_steps=2, second_order=1, hvp="exact", hvp_terms=hvp_terms)
    ad, _ = make_adaptor(opts, False)
    n = int(gso["nframes"])
    ad.reset_records(n)
    hmr = ad.model.module
    theta0 = hmr.theta.detach().clone()
    names = [str(x) for x in gso["names"]]
    for step in range(n):
        ad.global_step = step
        ad.fit_losses = {}
        batch = {k: v.to(ad.device) for k, v in assets.make_frame(step, 1, seed=22).items()}
        ad.model.eval()
        ad.adaptation(batch)
        up = float(ad.fit_losses["ul/total"])
        assert abs(up - gso["upper_loss"][step]) < 1e-4 * abs(gso["upper_loss"][step]), (step, up, gso["upper_loss"][step])
        assert ad.optim_step_record[-1] == int(gso["extra_steps"][step])
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
        for k, v in dict(rotmat=r, shape=s, cam=c).items():
            assert rel_err(v.cpu().numpy(), gso[f"pred{step}_{k}"]) < 1e-3, (step, k)
        if step == 0:
            st = ad.optimizer.state[hmr.theta]
            g1 = hmr._layout1.unpack((st["exp_avg"] / (1 - ad.options.beta1)).cpu())
            gn = np.array([float(g1[k].double().norm()) for k in names])
            err = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
            gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
            sl = {k[3:]: cosine(g1[k[3:]].flatten()[:256].numpy(), gso[k]) for k in gso.files if k.startswith("g1_") and k != "g1_norms"}
            print("SO full set, hvp_terms=%s: grad-norm error median %.2e max %.2e (FO-SO gap median %.2e), min slice cosine %.6f" % (
                hvp_terms, np.median(err), err.max(), np.median(gap), min(sl.values())))
            if hvp_terms == "all":
                assert np.median(err) < 5e-4 and err.max() < 5e-3, (np.median(err), err.max())
                assert min(sl.values()) > 0.999, sl
            else:
                assert np.median(err) < 0.1 * np.median(gap) and err.max() < 6e-2, (np.median(err), err.max(), np.median(gap))
    # element-wise slices only for the exact products: the difference quotient of a whole level is noisy element by element in the
    # backbone (measured on MI355X after 4 frames: cosine 0.966 on the layer2.0.conv2 slice of Adam's m, norms within the 2 % bound)
    assert_final_state_matches_golden(ad, gso, theta0, opts, slices=hvp_terms == "all")


@pytest.mark.parametrize("overlap", [0, 1])
def test_native_stepper_is_bit_identical_to_autograd_path(overlap):
    """One C call per frame (csrc/adapt_step.hip, the default for the frame-loss configurations) against the
    torch.autograd composition of the same kernels on the 3-inner-step stream, with and without the side stream for
    the final inference: identical weights / Adam state, metric records equal to rounding."""
    from dynaboa_amd import assets
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(4)]
    opts, ident = STREAMS["fo_inner3_frameonly"]
    outs = []
    for native in (1, 0):
        ad, _ = make_adaptor(dict(opts, native_step=native, overlap_metrics=overlap), ident, deferred=1)
        res = ad.excute(frames, nframes=4)
        assert (ad._native is not None) == bool(native)
        st = ad.optimizer.state[ad.model.module.theta]
        recs = [(r["step"], r["tag"], float(np.ravel(r["mpjpe"])[0]), float(np.ravel(r["pampjpe"])[0]), r["pve"]) for r in ad.metric_records]
        outs.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"], res, recs,
                     {k: float(v) for k, v in ad.last_summaries.items()}))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3] == 4
    assert len(a[5]) == len(b[5]) == 16
    for x, y in zip(a[5], b[5]):
        assert x[:2] == y[:2]
        np.testing.assert_allclose(x[2:], y[2:], rtol=2e-5)
    for k in a[6]:
        assert abs(a[6][k] - b[6][k]) <= 1e-6 * abs(b[6][k]), k


def test_native_stepper_immediate_metrics_and_batch():
    """deferred_metrics=0 (values returned per frame like the reference) and batch 2 through the native stepper."""
    from dynaboa_amd import assets
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 2, seed=22).items()} for s in range(2)]
    outs = []
    for native in (1, 0):
        ad, _ = make_adaptor(dict(FRAME_ONLY, inner_step=2, batch_size=2, native_step=native), False, deferred=0)
        res = ad.excute(frames, nframes=2)
        outs.append((ad.model.module.theta.detach().clone(), res, ad.mpjpe_all_lower))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in ("mpjpe", "pampjpe", "pve"):
        np.testing.assert_allclose(np.ravel(np.array(outs[0][1][k], np.float64)), np.ravel(np.array(outs[1][1][k], np.float64)), rtol=2e-5)
    np.testing.assert_allclose(np.ravel(np.array(outs[0][2], np.float64)), np.ravel(np.array(outs[1][2], np.float64)), rtol=2e-5)


def test_replica_group_is_bit_identical_to_single_sequences():
    """S = 3 independent sequences stepped in lockstep by one native stepper (every launch covers all replicas;
    csrc/dyb_common.h) against the same three sequences adapted one at a time: per replica identical weights / Adam state,
    metric records equal to rounding.  Different checkpoints and frames per replica, so a pointer left on replica 0's
    arena would show."""
    from dynaboa_amd import assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    S, NF = 3, 3

    def mk(r):
        o = DB.frame_only_options(inner_step=3)
        o.deferred_metrics = 1
        return DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True), device="cuda:0")
    frames = [[{k: v.to("cuda:0") for k, v in assets.make_frame(100 * r + s, 1, seed=22).items()} for s in range(NF)] for r in range(S)]
    singles = []
    for r in range(S):
        ad = mk(r)
        res = ad.excute(frames[r], nframes=NF)
        st = ad.optimizer.state[ad.model.module.theta]
        singles.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), res))
    ads = [mk(r) for r in range(S)]
    grp = NS.ReplicaGroup(ads, NF)
    for s in range(NF):
        grp.step([frames[r][s] for r in range(S)], s)
    fl = grp.flush_metrics()
    for r in range(S):
        a = ads[r]
        st = a.optimizer.state[a.model.module.theta]
        assert st["step"] == NF
        assert torch.equal(a.model.module.theta.detach(), singles[r][0]), r
        assert torch.equal(st["exp_avg"], singles[r][1]) and torch.equal(st["exp_avg_sq"], singles[r][2]), r
        for k in ("mpjpe", "pampjpe", "pve"):
            np.testing.assert_allclose(np.ravel(np.array(fl[r][k], np.float64)), np.ravel(np.array(singles[r][3][k], np.float64)), rtol=2e-5)
    # the replicas really differ from each other
    assert not torch.equal(ads[0].model.module.theta.detach(), ads[1].model.module.theta.detach())


def test_train_mode_dropout_matches_reference_distribution(ckpt_rand):
    """HMR.train() against the reference module left in train() mode (golden g8: 512 forwards of one frame with live
    nn.Dropout(0.5), what the reference's mean teacher does - base_adaptor.py:151-158).  RNG streams cannot be matched, the
    distribution can: the regressor is linear downstream of the masks, so the mean equals the eval output and the
    per-output standard deviation is fixed by the weights."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr import hmr
    g = golden("g8_dropout.npz")
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    m = hmr(mp, seed=1).to("cuda:0")
    m.load_state_dict(ckpt_rand, strict=True)
    img = assets.make_frame(0, 1, seed=22)["image"].to("cuda:0")
    N = 1024
    m.train()
    torch.manual_seed(5)
    with torch.no_grad():
        o = torch.stack([torch.cat([x.flatten() for x in m(img)]) for _ in range(N)]).double().cpu()
    mean, std = o.mean(0).numpy(), o.std(0).numpy()
    ref_mean, ref_std = g["mean"], g["std"]
    big = ref_std > 0.2 * ref_std.max()
    assert big.sum() > 20
    ratio = std[big] / ref_std[big]
    print("dropout std ratio: median %.3f min %.3f max %.3f" % (np.median(ratio), ratio.min(), ratio.max()))
    assert 0.9 < np.median(ratio) < 1.1 and ratio.min() > 0.75 and ratio.max() < 1.3        # two Monte-Carlo estimates (512 / 1024 draws)
    se = np.sqrt(ref_std ** 2 / int(g["n"]) + std ** 2 / N)
    assert (np.abs(mean - ref_mean)[big] < 6 * se[big]).all()
    m.eval()
    with torch.no_grad():
        e = torch.cat([x.flatten() for x in m(img)]).cpu().numpy()
    assert rel_err(e, g["eval"]) < 1e-4


def test_teacher_dropout_option_runs_reference_teacher_mode():
    """--teacher_dropout 1: the mean teacher stays in train() mode like the reference's; the teacher term becomes noisy
    (two identical runs with different seeds disagree in it) while --teacher_dropout 0 is deterministic."""
    from dynaboa_amd import assets
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(2)]
    vals = {}
    for td in (0, 1):
        for seed in (1, 2):
            ad, _ = make_adaptor(dict(STREAMS["fo_inner1_full"][0], teacher_dropout=td, dynamic_boa=0), False)
            torch.manual_seed(seed)
            ad.excute(frames, nframes=2)
            assert ad.teacher.training == bool(td)
            vals[(td, seed)] = float(ad.last_summaries["teacher/loss"])
    assert vals[(0, 1)] == vals[(0, 2)]
    assert vals[(1, 1)] != vals[(1, 2)] and np.isfinite(vals[(1, 1)]) and vals[(1, 1)] > vals[(0, 1)]


def test_train_mode_teacher_on_the_native_stepper_equals_the_autograd_path():
    """VERDICT r5 missing 8: the reference never calls teacher.eval() (base_adaptor.py:151-158), so its teacher forwards run with live
    nn.Dropout(0.5) after fc1 / fc2 (model/hmr.py:84,86).  --teacher_dropout 1 now stays on the native stepper: every teacher forward of a
    frame (one per upper level, dynamic-loop steps included) draws its masks from the keys the autograd path would draw - torch's seed and the
    process-wide train-forward counter - so the two paths see the SAME masks: over a stream whose dynamic loop is entered they agree as
    closely as they do with the eval-mode teacher (test_native_full_term_set_matches_autograd_path: the passes' gradients are summed in
    a different order, nothing else), the first frame's teacher term to 1e-6, and the run differs from the eval-mode teacher's."""
    from dynaboa_amd import assets, hmr as H
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(4)]
    opts = dict(STREAMS["fo_inner1_full"][0], teacher_dropout=1, cos_sim_threshold=1.0e-4, optim_steps=3)
    outs = {}
    for native in (1, 0, 1):
        torch.manual_seed(5)
        H._DROP_CALLS[0] = 0
        ad, _ = make_adaptor(dict(opts, native_step=native), False)
        ad.reset_records(4)
        tl = []
        for s_, fr in enumerate(frames):
            ad.global_step = s_
            ad.fit_losses = {}
            ad.model.eval()
            ad.adaptation(fr)
            tl.append(float(ad.fit_losses["teacher/loss"]))
        ad.flush_metrics()
        assert (ad._native is not None and ad._native.full) == bool(native) and ad.teacher.training
        st = ad.optimizer.state[ad.model.module.theta]
        res = (ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), ad.teacher.theta.detach().clone(), list(ad.optim_step_record),
               H._DROP_CALLS[0], tl)
        if native in outs:
            for a, b in zip(outs[native], res):
                assert torch.equal(a, b) if torch.is_tensor(a) else a == b          # same seed, same counter: reproducible
        outs[native] = res
    assert outs[1][4] == outs[0][4] >= 4 + sum(min(x, 3) for x in outs[1][3])       # one key per teacher forward, on both paths
    assert outs[1][3] == outs[0][3] and max(outs[1][3]) >= 1, (outs[1][3], outs[0][3])
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    print("train-mode teacher, native vs autograd: theta %.2e m %.2e teacher %.2e; teacher term per frame %s vs %s" %
          (rel(outs[1][0], outs[0][0]), rel(outs[1][1], outs[0][1]), rel(outs[1][2], outs[0][2]), outs[1][5], outs[0][5]))
    assert abs(outs[1][5][0] - outs[0][5][0]) < 1e-6 * abs(outs[0][5][0])            # same masks: frame 0's teacher term is the same number
    np.testing.assert_allclose(outs[1][5], outs[0][5], rtol=1e-3)          # (3e-4 measured at frame 3: the weights differ by ~2e-6 by then)
    assert rel(outs[1][0], outs[0][0]) < 5e-6 and rel(outs[1][2], outs[0][2]) < 5e-6 and rel(outs[1][1], outs[0][1]) < 5e-3
    torch.manual_seed(5)
    ad, _ = make_adaptor(dict(opts, teacher_dropout=0), False)
    ad.excute(frames, nframes=4)
    assert not torch.equal(ad.model.module.theta.detach(), outs[1][0])               # the eval-mode teacher gives another run (4.9e-6 apart after 4 frames)
    assert abs(float(ad.last_summaries["teacher/loss"]) - outs[1][5][-1]) > 1e-3 * outs[1][5][-1]


def test_bf16_mfma_variant_vs_fp32_batch16(ckpt_rand):
    """BASELINE configs[4], fp32-vs-bf16 arm: the engine at batch 16 with the convolutions on the bf16 matrix cores (fp32
    master weights / activations / statistics / accumulators, operand tiles rounded as they are staged) against the exact
    fp32 engine - outputs within 1e-2, parameter gradients aligned (cosine > 0.99 per tensor); then two adapted frames:
    the weight update is not lost to rounding (fp32 master weights: per-tensor delta norms follow the fp32 run)."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr import get_layout, hmr
    B = 16
    L = get_layout(B)
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    m = hmr(mp, seed=1).to("cuda:0").eval()
    m.load_state_dict(ckpt_rand, strict=True)
    img = assets.make_frame(0, B, seed=22)["image"].to("cuda:0")
    g = torch.Generator().manual_seed(7)
    wr, ws_, wc = (torch.randn(s, generator=g).to("cuda:0") for s in ((B, 24, 3, 3), (B, 10), (B, 3)))
    res = {}
    try:
        for mode in (0, 1):
            L.set_bf16(bool(mode))
            m.theta.grad = None
            r, s, c = m(img)
            ((r * wr).sum() + (s * ws_).sum() + (c * wc).sum()).backward()
            res[mode] = (r.detach().clone(), s.detach().clone(), c.detach().clone(), m._layout1.unpack(m.theta.grad))
    finally:
        L.set_bf16(False)
    for i, name in enumerate(("rotmat", "shape", "cam")):
        e = rel_err(res[1][i].cpu().numpy(), res[0][i].cpu().numpy())
        assert 0 < e < 1e-2, (name, e)                     # different (bf16 operands) but close
    cos = {k: cosine(res[1][3][k].cpu().numpy(), res[0][3][k].cpu().numpy()) for k in res[0][3]}
    worst = min(cos, key=cos.get)
    print("bf16 vs fp32 gradient cosine: min %.5f (%s), median %.5f" % (cos[worst], worst, float(np.median(list(cos.values())))))
    # measured on MI355X with this random-initialised network: 1.0000 in the regressor, > 0.999 in layer4, falling towards
    # the stem as the rounding noise of ~100 bf16 products in a row accumulates (0.944 at conv1; median 0.998) - DESIGN.md 4
    assert float(np.median(list(cos.values()))) > 0.995 and cos[worst] > 0.9
    for k, v in cos.items():
        if k.startswith(("layer4", "fc", "dec")):
            assert v > 0.999, (k, v)
        elif k.startswith("layer3"):
            assert v > 0.99, (k, v)
    # adaptation: same stream in both modes
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(2)]
    deltas = {}
    try:
        for mode in (0, 1):
            ad, _ = make_adaptor(dict(FRAME_ONLY, inner_step=2, bf16_mfma=mode), False, deferred=1)
            assert get_layout(1).bf16 == bool(mode)
            t0 = ad.model.module.theta.detach().clone()
            ad.excute(frames, nframes=2)
            d = ad.model.module._layout1.unpack((ad.model.module.theta.detach().double() - t0.double()).float())
            deltas[mode] = {k: float(v.double().norm()) for k, v in d.items()}
    finally:
        get_layout(1).set_bf16(False)
    ratio = np.array([deltas[1][k] / deltas[0][k] for k in deltas[0] if deltas[0][k] > 0])
    assert 0.9 < np.median(ratio) < 1.1 and ratio.min() > 0.5, (np.median(ratio), ratio.min())


def test_engine_batch16_dispatches_agree(ckpt_rand):
    """Regression: at batch 16 the single-launch 1x1 data gradients leave MORE partial records per layer (25 row tiles per
    28x28 image) than the reduce kernel's own layout (16 chunks per image); the per-layer slots are sized for both.  Both
    dispatches must give the same parameter gradients (they once did not: a tile-layout block overran its slot and
    corrupted the next layer's GroupNorm weight / bias gradients)."""
    from dynaboa_amd import _lib, assets
    from dynaboa_amd.hmr import hmr
    lib = _lib.load()
    B = 16
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    m = hmr(mp, seed=1).to("cuda:0").eval()
    m.load_state_dict(ckpt_rand, strict=True)
    img = assets.make_frame(0, B, seed=22)["image"].to("cuda:0")
    g = torch.Generator().manual_seed(7)
    wr, ws_, wc = (torch.randn(s, generator=g).to("cuda:0") for s in ((B, 24, 3, 3), (B, 10), (B, 3)))
    res = {}
    try:
        for k4 in (1, 0):
            lib.dyb_set_option(b"k4", k4); lib.dyb_set_option(b"k4_bwd", k4)
            m.theta.grad = None
            r, s, c = m(img)
            ((r * wr).sum() + (s * ws_).sum() + (c * wc).sum()).backward()
            res[k4] = m._layout1.unpack(m.theta.grad)
    finally:
        lib.dyb_set_option(b"k4", 1); lib.dyb_set_option(b"k4_bwd", 1)
    for k in res[1]:
        a, b = res[1][k].double().flatten(), res[0][k].double().flatten()
        assert cosine(a.cpu().numpy(), b.cpu().numpy()) > 0.9999 and abs(float(a.norm() / b.norm()) - 1) < 2e-3, k


@pytest.mark.parametrize("tag", ["fo_inner1_full", "fo_inner1_full_forced"])
def test_native_full_term_set_matches_autograd_path(tag):
    """The reference's full term set (teacher + motion + labelled exemplars + dynamic loop) through the native stepper
    (dyb_stepper_adapt_frame_full: one C call per frame, loss heads in dyb_aux_loss_terms, gate cosines polled from pinned
    memory) against the torch.autograd composition: same extra-step counts, weights / Adam moments / teacher equal to
    rounding (the passes' gradients are summed in a different order), every logged term equal to 1e-5."""
    from dynaboa_amd import assets
    opts, ident = STREAMS[tag]
    frames = [{k: v.to("cuda:0") for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(4)]
    outs = []
    for native in (1, 0):
        ad, _ = make_adaptor(dict(opts, native_step=native), ident, deferred=1)
        res = ad.excute(frames, nframes=4)
        assert (ad._native is not None and ad._native.full) == bool(native)
        st = ad.optimizer.state[ad.model.module.theta]
        outs.append(dict(theta=ad.model.module.theta.detach().clone(), m=st["exp_avg"].clone(), v=st["exp_avg_sq"].clone(), t=st["step"],
                         teacher=ad.teacher.theta.detach().clone(), res=res, steps=list(ad.optim_step_record),
                         log={k: float(v) for k, v in ad.last_summaries.items()}, recs=[(r["step"], r["tag"]) for r in ad.metric_records],
                         sims=[[float(d[12]["cos"]) for d in ad.feat_sims[s]] for s in sorted(ad.feat_sims)]))
    a, b = outs
    assert a["steps"] == b["steps"] and a["t"] == b["t"] and a["recs"] == b["recs"]
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    L = ad.model.module._layout1
    ma, mb = L.unpack(a["m"]), L.unpack(b["m"])
    per = {k: rel(ma[k], mb[k]) for k in ma}
    worst = sorted(per.items(), key=lambda kv: -kv[1])[:6]
    print("native vs autograd (%s): theta %.2e teacher %.2e m %.2e v %.2e; worst per-tensor m: %s" %
          (tag, rel(a["theta"], b["theta"]), rel(a["teacher"], b["teacher"]), rel(a["m"], b["m"]), rel(a["v"], b["v"]),
           [(k, "%.1e" % v) for k, v in worst]))
    # weights move by ~1e-4 relative per Adam step: 5e-6 after up to 12 steps = the updates agree to a few per cent of one step
    assert rel(a["theta"], b["theta"]) < 5e-6 and rel(a["teacher"], b["teacher"]) < 5e-6
    assert rel(a["m"], b["m"]) < 5e-3 and rel(a["v"], b["v"]) < 5e-3              # ReLU-flip noise class
    for k in ("mpjpe", "pampjpe", "pve"):
        np.testing.assert_allclose(np.ravel(np.array(a["res"][k], np.float64)), np.ravel(np.array(b["res"][k], np.float64)), rtol=2e-5)
    assert a["log"].keys() == b["log"].keys()
    for k in a["log"]:
        assert abs(a["log"][k] - b["log"][k]) <= 2e-4 * abs(b["log"][k]) + 1e-9, k       # last frame's terms: the weights differ by ~2e-6 by then
    np.testing.assert_allclose(np.array(a["sims"]), np.array(b["sims"]), rtol=0, atol=1e-6)


def test_replica_group_with_replica_aware_split(restore_switches):
    """rep_split=1 (what bench.py uses for several sequences per GPU): the split-K depth is chosen for the replica-multiplied
    grid, so summation order - not arithmetic - differs from a sequence running alone: weights equal to fp32 rounding."""
    from dynaboa_amd import _lib, assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    S, NF = 4, 2

    def mk(r):
        o = DB.frame_only_options(inner_step=3)
        o.deferred_metrics = 1
        return DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True), device="cuda:0")
    frames = [[{k: v.to("cuda:0") for k, v in assets.make_frame(100 * r + s, 1, seed=22).items()} for s in range(NF)] for r in range(S)]
    singles = []
    for r in range(S):
        ad = mk(r)
        ad.excute(frames[r], nframes=NF)
        singles.append((ad.model.module.theta.detach().clone(), ad.optimizer.state[ad.model.module.theta]["exp_avg"].clone()))
    _lib.load().dyb_set_option(b"rep_split", 1)
    ads = [mk(r) for r in range(S)]
    grp = NS.ReplicaGroup(ads, NF)
    for s in range(NF):
        grp.step([frames[r][s] for r in range(S)], s)
    grp.flush_metrics()
    _lib.load().dyb_set_option(b"rep_split", 0)
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    for r in range(S):
        a = ads[r]
        assert rel(a.model.module.theta.detach(), singles[r][0]) < 1e-6, r
        assert rel(a.optimizer.state[a.model.module.theta]["exp_avg"], singles[r][1]) < 5e-3, r


def test_configs4_batch16_first_vs_second_order_and_bf16_vs_oracle(gmm_t, smpl_tabs):
    """BASELINE configs[4] ("first-order vs second-order outer grad ablation at batch=16, fp32 vs bf16 MFMA") against the CPU
    ORACLE: one bilevel frame at batch 16, inner_step 1, frame losses -
      * fp32, first order and second order (exact Hessian-vector products): outer gradient vs the oracle's first_order=True /
        create_graph=True gradient (norm 2e-2, cosine 0.999 on sampled tensors); the two gradients really differ;
      * bf16 MFMA, first order: predictions within 1e-2 of the oracle, outer-gradient cosine > 0.99 from layer3 up, > 0.9 everywhere
        sampled (operands rounded to bf16, fp32 accumulate / master weights / statistics)."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr import get_layout
    from oracle import ref_cpu as O
    B = 16
    frame = assets.make_frame(5, B, seed=22)
    names = ["conv1.weight", "layer2.0.conv2.weight", "layer3.5.conv1.weight", "layer4.2.conv3.weight", "fc1.weight", "decpose.weight"]
    torch.set_num_threads(32)
    sd, ref = None, {}
    ours = {}
    try:
        for tag, over in (("fo", {}), ("so", dict(second_order=1)), ("bf16", dict(bf16_mfma=1))):
            opts = dict(FRAME_ONLY, inner_step=1, batch_size=B, **over)
            ad, bundle = make_adaptor(opts, False)
            ad.reset_records(1)
            ad.global_step = 0
            ad.model.eval()
            ad.adaptation({k: v.to(ad.device) for k, v in frame.items()})
            hmr_m = ad.model.module
            g = hmr_m._layout1.unpack(ad.optimizer.state[hmr_m.theta]["exp_avg"] / (1 - ad.options.beta1))
            with torch.no_grad():
                r, s, c = ad.model(frame["image"].to(ad.device))
            ours[tag] = dict(g={k: g[k].cpu() for k in names}, pred=dict(rotmat=r.cpu(), shape=s.cpu(), cam=c.cpu()))
            if sd is None:
                sd = {k.replace("module.", ""): v for k, v in bundle.checkpoint["model"].items()}
            get_layout(B).set_bf16(False)
            del ad
    finally:
        get_layout(B).set_bf16(False)
    for fo in (True, False):
        o = O.Adapter(sd, O.smpl_tables_to_torch(smpl_tabs), gmm_t, dict(FRAME_ONLY, inner_step=1, batch_size=B), first_order=fo)
        ref["fo" if fo else "so"] = o.adapt_frame(frame)
    for tag in ("fo", "so"):
        for k in names:
            a, b = ours[tag]["g"][k].double().flatten(), ref[tag]["outer_grad"][k].double().flatten()
            assert cosine(a, b) > 0.999, (tag, k, cosine(a, b))
            assert abs(float(a.norm() / b.norm()) - 1) < 2e-2, (tag, k)
    gap = max(rel_err(ref["fo"]["outer_grad"][k].numpy(), ref["so"]["outer_grad"][k].numpy()) for k in names)
    assert gap > 1e-2, gap                                      # first and second order are different gradients here
    for k, v in ours["bf16"]["pred"].items():
        assert rel_err(v.numpy(), ref["fo"]["pred"][k].numpy()) < 1e-2, k
    for k in names:
        cs = cosine(ours["bf16"]["g"][k].double().flatten(), ref["fo"]["outer_grad"][k].double().flatten())
        assert cs > (0.99 if k.startswith(("layer3", "layer4", "fc", "dec")) else 0.9), (k, cs)
