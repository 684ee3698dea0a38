"""Sequence replicas beyond the frame-loss configuration on cuda:0:
  * the reference's DEFAULT term set (teacher + motion + labelled exemplars + dynamic-BOA loop, dynaboa_benchmark.py:126-193) for a
    ReplicaGroup: every launch covers all sequences, the gate is decided per sequence (a converged one leaves the launch set), so the
    sequences take different numbers of Adam steps - each must end exactly where it ends when adapted alone;
  * sequences of different lengths (boa_dataset/pw3d.py:19-35): a sequence whose stream has ended leaves the active set."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FULL = dict(inner_step=1, interval=2, optim_steps=2)


def _mk(r, opts):
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    o = DB.parser.parse_args([])
    for k, v in opts.items():
        setattr(o, k, v)
    o.deferred_metrics = 1
    return DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True, smpl_seed=0), device="cuda:0")


def _frames(S, NF):
    from dynaboa_amd import assets
    return [[{k: v.to("cuda:0") for k, v in assets.make_frame(100 * r + s, 1, seed=22).items()} for s in range(NF)] for r in range(S)]


def _state(ad):
    st = ad.optimizer.state[ad.model.module.theta]
    return dict(theta=ad.model.module.theta.detach().clone(), m=st["exp_avg"].clone(), v=st["exp_avg_sq"].clone(), t=int(st["step"]),
                teacher=ad.teacher.theta.detach().clone() if getattr(ad, "teacher", None) is not None else None,
                steps=list(ad.optim_step_record))


def test_full_term_replica_group_matches_single_sequences():
    from dynaboa_amd import native_step as NS
    S, NF = 3, 4
    frames = _frames(S, NF)
    # 1. a threshold that separates the sequences at the first gate: run them alone with the loop forced on and read 1 - cos_12
    gaps = []
    for r in range(S):
        ad = _mk(r, dict(FULL, cos_sim_threshold=-1.0))
        ad.excute(frames[r][:1], nframes=1)
        gaps.append(1.0 - float(ad.feat_sims[0][0][12]["cos"]))
    thr = float(np.sort(gaps)[S // 2] * 0.999)            # the sequences at / above the median go on, those below stop
    opts = dict(FULL, cos_sim_threshold=thr)
    # 2. alone
    singles, single_res = [], []
    for r in range(S):
        ad = _mk(r, opts)
        single_res.append(ad.excute(frames[r], nframes=NF))
        assert ad._native is not None and ad._native.full
        singles.append(_state(ad))
    assert len({tuple(s["steps"]) for s in singles}) > 1, [s["steps"] for s in singles]      # the sequences really took different paths
    # 3. in lockstep
    ads = [_mk(r, opts) for r in range(S)]
    grp = NS.ReplicaGroup(ads, NF)
    assert grp.stepper.full and grp.stepper.S == S
    for s in range(NF):
        grp.step([frames[r][s] for r in range(S)], s)
    fl = grp.flush_metrics()
    for r in range(S):
        g, a = _state(ads[r]), singles[r]
        assert g["steps"] == a["steps"], (r, g["steps"], a["steps"])
        assert g["t"] == a["t"], (r, g["t"], a["t"])
        for k in ("theta", "m", "v", "teacher"):
            assert torch.equal(g[k], a[k]), (r, k, float((g[k].double() - a[k].double()).norm() / a[k].double().norm()))
        for k in ("mpjpe", "pampjpe", "pve"):
            np.testing.assert_allclose(np.ravel(np.array(fl[r][k], np.float64)), np.ravel(np.array(single_res[r][k], np.float64)), rtol=2e-5)
    assert not torch.equal(ads[0].model.module.theta.detach(), ads[1].model.module.theta.detach())


def test_full_term_replica_group_throughput_schedule():
    """The same at S = 8 with the replica-aware policy (rep_split = 1: every launch on the throughput schedule, what bench.py's
    default-flags record uses): summation order differs from a sequence running alone, arithmetic does not."""
    from dynaboa_amd import _lib, native_step as NS
    S, NF = 8, 2
    frames = _frames(S, NF)
    opts = dict(FULL, cos_sim_threshold=-1.0)             # every sequence takes optim_steps extra steps: the counts cannot flip on rounding
    singles = []
    for r in range(S):
        ad = _mk(r, opts)
        ad.excute(frames[r], nframes=NF)
        singles.append(_state(ad))
    lib = _lib.load()
    lib.dyb_set_option(b"rep_split", 1)
    try:
        ads = [_mk(r, opts) for r in range(S)]
        grp = NS.ReplicaGroup(ads, NF)
        for s in range(NF):
            grp.step([frames[r][s] for r in range(S)], s)
        grp.flush_metrics()
    finally:
        lib.dyb_set_option(b"rep_split", 0)
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    for r in range(S):
        g, a = _state(ads[r]), singles[r]
        assert g["steps"] == a["steps"] and g["t"] == a["t"], r
        assert rel(g["theta"], a["theta"]) < 5e-6 and rel(g["teacher"], a["teacher"]) < 5e-6, r
        assert rel(g["m"], a["m"]) < 5e-3 and rel(g["v"], a["v"]) < 1e-2, r


def test_ragged_sequences_leave_the_active_set():
    """Three sequences of 3, 2 and 1 frames in one group (frame-loss configuration): each ends where it ends alone, with its own
    number of Adam steps."""
    from dynaboa_amd import benchmark as DB, native_step as NS
    lens = [3, 2, 1]
    S = len(lens)
    frames = _frames(S, max(lens))
    fo = {k: v for k, v in vars(DB.frame_only_options(inner_step=3)).items()}
    singles, single_res = [], []
    for r in range(S):
        ad = _mk(r, fo)
        single_res.append(ad.excute(frames[r][:lens[r]], nframes=lens[r]))
        singles.append(_state(ad))
    ads = [_mk(r, fo) for r in range(S)]
    grp = NS.ReplicaGroup(ads, max(lens))
    for s in range(max(lens)):
        out = grp.step([frames[r][s] if s < lens[r] else None for r in range(S)], s)
        assert [o is not None for o in out] == [s < lens[r] for r in range(S)]
    fl = grp.flush_metrics()
    for r in range(S):
        g, a = _state(ads[r]), singles[r]
        assert g["t"] == a["t"] == lens[r], (r, g["t"])
        for k in ("theta", "m", "v"):
            assert torch.equal(g[k], a[k]), (r, k)
        assert len(fl[r]["mpjpe"]) == lens[r]
        np.testing.assert_allclose(np.ravel(np.array(fl[r]["mpjpe"], np.float64)), np.ravel(np.array(single_res[r]["mpjpe"], np.float64)), rtol=2e-5)


def test_sharded_driver_on_one_gpu_matches_sequences_alone():
    """dynaboa_amd.sharded.run_sharded over five synthetic sequences of 3, 1, 2, 2, 1 frames as two shards (run one after the other
    on this GPU), two sequences at a time per shard: every frame of the stream comes back once, with the errors the sequence gives
    when adapted alone."""
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.sharded import SequenceSpec, run_sharded
    lens = [3, 1, 2, 2, 1]
    frames = _frames(len(lens), max(lens))
    fo = {k: v for k, v in vars(DB.frame_only_options(inner_step=3)).items()}
    specs, first = [], 0
    for k, n in enumerate(lens):
        specs.append(SequenceSpec(f"s{k}", first, n, (lambda k=k, n=n: frames[k][:n])))
        first += n
    alone = []
    for k, n in enumerate(lens):
        ad = _mk(0, fo)                                   # every sequence starts from the same checkpoint
        res = ad.excute(frames[k][:n], nframes=n)
        alone += [float(np.ravel(x)[0]) for x in res["mpjpe"]]
    got = {}
    for rank in (0, 1):
        res = run_sharded(DB.frame_only_options(inner_step=3), specs, lambda: _mk(0, fo), num_shards=2, shard_rank=rank, seqs_per_gpu=2)
        assert res["frames_local"] == sum(lens[i] for i in res["owned"])
        for gi, m in zip(res["global_index"], res["mpjpe"]):
            assert int(gi) not in got
            got[int(gi)] = float(m)
    assert sorted(got) == list(range(sum(lens)))
    np.testing.assert_allclose([got[i] for i in range(sum(lens))], alone, rtol=2e-5)


def test_parallel_passes_of_one_sequence_are_bit_identical_to_the_sequential_level(monkeypatch):
    """One sequence, default term set, the dynamic loop entered (a low threshold, at most three extra steps): the history pass and the
    exemplar pass on the stepper's own streams (par_passes 1, round 5) against the sequential level (0) - weights, Adam moments,
    teacher, extra-step counts and metrics bit for bit over four frames (interval 2: the last two have a history frame)."""
    frames = _frames(1, 4)[0]
    outs = []
    for par in ("0", "1", "1"):
        monkeypatch.setenv("DYB_PAR_PASSES", par)           # read when the stepper is created
        ad = _mk(0, dict(FULL, cos_sim_threshold=1.0e-4, optim_steps=3))
        res = ad.excute(frames, nframes=4)
        assert ad._native is not None and ad._native.full
        st = _state(ad)
        outs.append([st["theta"], st["m"], st["v"], st["teacher"], torch.tensor(st["steps"]),
                     torch.tensor(np.ravel(np.array(res["mpjpe"], np.float64)))])
        del ad
    assert len(outs[0][4]) == 4 and int(outs[0][4].max()) >= 1, outs[0][4]       # the loop was entered
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_forward_shared_across_dynamic_loop_steps_is_bit_identical_to_the_literal_two_forwards(monkeypatch):
    """an extra step of the dynamic-BOA loop reuses the previous step's final inference as its upper level's forward
    (share_dyn_fwd 1, the default) - against the literal schedule (0: the upper level runs its own forward,
    dynaboa_benchmark.py:170-181) weights, Adam moments, teacher, step counts and metrics must agree bit for bit, and
    dyb_stepper_output must hand out the LAST final inference whichever arena it ended in (odd and even step counts occur)."""
    frames = _frames(1, 4)[0]
    outs = []
    for share in ("0", "1"):
        monkeypatch.setenv("DYB_SHARE_DYN_FWD", share)      # read when the stepper is created
        ad = _mk(0, dict(FULL, cos_sim_threshold=1.0e-4, optim_steps=3))
        res = ad.excute(frames, nframes=4)
        assert ad._native is not None and ad._native.full
        st = _state(ad)
        final = [ad._native.output(w).clone() for w in range(4)] if hasattr(ad._native, "output") else []
        outs.append([st["theta"], st["m"], st["v"], st["teacher"], torch.tensor(st["steps"]),
                     torch.tensor(np.ravel(np.array(res["mpjpe"], np.float64)))] + final)
        del ad
    steps = outs[0][4]
    assert int(steps.max()) >= 1, steps
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["fo_inner1_full_gated", "fo_inner1_full_gated_b"])
def test_gated_reference_stream_as_replica_0_of_a_group_of_8(tag):
    """VERDICT r5 item 1, replica form: eight sequences with the reference's literal default flags in lockstep under the drivers'
    replica policy (rep_split 1: every launch on the throughput schedule, the gate decided per sequence, a converged sequence leaving
    the launch set).  Replica 0 carries the checkpoint and frames of the reference-generated golden `g5_<tag>` (tools/make_golden.py
    g5_gated: the dynamic-BOA loop leaves by convergence / never opens / runs into the cut-off, every decision >= gate_margin away
    from the threshold) and must reproduce the reference's step counts, every check's cosines, and its final Adam state - with a
    different summation order than the reference's and than a sequence alone."""
    from conftest import golden
    from test_adaptation_gpu import assert_final_state_matches_golden, assert_gate_matches_golden
    from dynaboa_amd import assets, native_step as NS, _lib
    g = golden(f"g5_{tag}.npz")
    S, NF = 8, int(g["nframes"])
    opts = dict(inner_step=1, cos_sim_threshold=float(g["gate_threshold"]))
    frames = [[{k: v.to("cuda:0") for k, v in assets.make_frame((100 * r + s) if r else s, 1, seed=22).items()} for s in range(NF)]
              for r in range(S)]
    lib = _lib.load()
    NS.set_replica_policy(True)
    try:
        ads = [_mk(r, opts) for r in range(S)]
        assert ads[0].options.interval == 5 and ads[0].options.optim_steps == 7
        theta0 = ads[0].model.module.theta.detach().clone()
        grp = NS.ReplicaGroup(ads, NF)
        assert grp.stepper.full and grp.stepper.S == S
        worst = 0.0
        for s in range(NF):
            grp.step([frames[r][s] for r in range(S)], s)
            worst = max(worst, assert_gate_matches_golden(ads[0], g, s))
            up = float(ads[0].fit_losses["ul/total"])
            assert abs(up - g["upper_loss"][s]) < 1e-4 * abs(g["upper_loss"][s]), (s, up, g["upper_loss"][s])
        fl = grp.flush_metrics()
    finally:
        lib.dyb_set_option(b"rep_split", 0)
        lib.dyb_set_option(b"tp_min", 8)
    print("gate %s, replica 0 of 8: worst |d(1 - cos12)| as a fraction of the check's distance from the threshold %.3f (margin %.2e)" % (tag, worst, float(g["gate_margin"])))
    steps = [int(x) for x in g["extra_steps"]]
    assert list(ads[0].optim_step_record) == steps
    assert len({tuple(a.optim_step_record) for a in ads}) > 1           # the sequences of the group took different paths
    for s in range(NF):
        assert abs(float(np.mean(fl[0]["mpjpe"][s])) - g["mpjpe"][s]) < 1e-3 * g["mpjpe"][s]
        assert abs(float(np.mean(fl[0]["pampjpe"][s])) - g["pampjpe"][s]) < 2e-3 * g["pampjpe"][s]
    assert_final_state_matches_golden(ads[0], g, theta0, dict(inner_step=1), tag=tag)


@pytest.mark.parametrize("S", [1, 4])
def test_teacher_ema_inside_the_adam_pass_is_bit_identical(S, monkeypatch):
    """"fuse_ema" (round 6): the mean teacher's EMA of an element is formed in the Adam pass that has just updated
    it (update_teacher follows optimizer.step(), dynaboa_benchmark.py:149-153) instead of a pass of its own re-reading theta - against
    the two-launch form: default term set, the dynamic loop entered, three frames, one sequence and a group of four (ranged updates):
    weights, Adam moments, teacher and step counts bit for bit."""
    from dynaboa_amd import native_step as NS
    NF = 3
    frames = _frames(S, NF)
    opts = dict(FULL, cos_sim_threshold=1.0e-4, optim_steps=3)
    outs = []
    for fe in ("0", "1"):
        monkeypatch.setenv("DYB_FUSE_EMA", fe)              # read when the stepper is created
        ads = [_mk(r, opts) for r in range(S)]
        if S == 1:
            ads[0].excute(frames[0], nframes=NF)
        else:
            grp = NS.ReplicaGroup(ads, NF)
            for s in range(NF):
                grp.step([frames[r][s] for r in range(S)], s)
            grp.flush_metrics()
        st = [_state(a) for a in ads]
        outs.append([torch.stack([x[k] for x in st]) for k in ("theta", "m", "v", "teacher")] + [torch.tensor([x["steps"] for x in st])])
        del ads
    assert int(outs[0][4].max()) >= 1
    for a, b in zip(*outs):
        assert torch.equal(a, b)
