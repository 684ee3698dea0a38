"""End-to-end parity of the BENCHMARKED configuration on cuda:0: a ReplicaGroup of S >= 8 sequences with
the replica-aware policy switched on exactly as bench.py's Runner does (`rep_split` = 1, default `tp_min` / `tp_gn_wgs` /
`tp_kernel`), i.e. every launch of the chain on the THROUGHPUT schedule (igemm_tp kernels, materialised dy, chunked GroupNorm).
Replica 0 carries the seed-22 checkpoint and frames of golden g5_fo_inner3_frameonly = the REFERENCE's own
Adaptor.adaptation() (dynaboa_benchmark.py:126-157) run frame after frame, and must meet the same assertions as
test_adaptation_gpu.py::test_stream_matches_reference; every replica must also match the same sequence adapted alone
(latency schedule)."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err
from test_adaptation_gpu import assert_final_state_matches_golden, assert_first_frame_outer_gradient

pytestmark = pytest.mark.gpu


@pytest.fixture
def headline_switches():
    """bench.py Runner.__init__ / the sharded driver for several sequences per GPU: native_step.set_replica_policy(True)."""
    from dynaboa_amd import _lib, native_step as NS
    lib = _lib.load()
    NS.set_replica_policy(True)          # rep_split = 1, throughput schedule from NS.TP_MIN_SEQUENCES (5) sequences per launch
    yield lib
    lib.dyb_set_option(b"rep_split", 0)
    lib.dyb_set_option(b"tp_min", 8)


def _mk(r):
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    o = DB.frame_only_options(inner_step=3)
    o.deferred_metrics = 1
    # replica 0 = the golden's checkpoint (tools/make_golden.py: seed 22, smpl_seed 0, randomised GroupNorm affine)
    return DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True, smpl_seed=0), device="cuda:0")


def _frames(S, NF):
    from dynaboa_amd import assets
    # replica 0 walks the golden's frames 0..NF-1; the others their own
    return [[{k: v.to("cuda:0") for k, v in assets.make_frame((100 * r + s) if r else s, 1, seed=22).items()} for s in range(NF)]
            for r in range(S)]


def _singles(S, NF, frames, lib):
    """every sequence adapted alone (one sequence per launch = latency schedule, whatever rep_split says)"""
    out = []
    for r in range(S):
        ad = _mk(r)
        res = ad.excute(frames[r], nframes=NF)
        st = ad.optimizer.state[ad.model.module.theta]
        out.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), res))
        del ad
    return out


def _rel(x, y):
    return float((x.double() - y.double()).norm() / y.double().norm())


def _opt(lib, name):
    import ctypes
    v = ctypes.c_int(-1)
    assert lib.dyb_get_option(name, ctypes.byref(v)) == 0, name
    return int(v.value)


def _throughput_schedule_is_on(lib, S):
    return _opt(lib, b"rep_split") == 1 and S >= _opt(lib, b"tp_min") and _opt(lib, b"tp_kernel") >= 1


@pytest.mark.parametrize("S", [8, 5])        # 5: the smallest group the drivers put on the throughput schedule (round 5: measured crossover)
def test_headline_schedule_stream_matches_reference_and_single_runs(S, headline_switches):
    lib = headline_switches
    assert _throughput_schedule_is_on(lib, S)
    from dynaboa_amd import native_step as NS
    g = golden("g5_fo_inner3_frameonly.npz")
    NF = int(g["nframes"])
    frames = _frames(S, NF)
    singles = _singles(S, NF, frames, lib)
    ads = [_mk(r) for r in range(S)]
    theta0 = ads[0].model.module.theta.detach().clone()
    grp = NS.ReplicaGroup(ads, NF)
    for step in range(NF):
        grp.step([frames[r][step] for r in range(S)], step)
        # replica 0 against the reference's run, frame by frame
        up = float(grp.stepper.losses(step, 3, 0)[3])
        assert abs(up - g["upper_loss"][step]) < 1e-4 * abs(g["upper_loss"][step]), (step, up, g["upper_loss"][step])
        a0 = ads[0]
        a0.model.eval()
        with torch.no_grad():
            r_, s_, c_ = a0.model(frames[0][step]["image"])
            j_ = a0.decode_smpl_params(r_, s_)["s3d"]
        for k, v in dict(rotmat=r_, shape=s_, cam=c_, joints=j_).items():
            assert rel_err(v.cpu().numpy(), g[f"pred{step}_{k}"]) < 1e-3, (step, k)     # north_star: 1e-3 rel
    fl = grp.flush_metrics()
    for step in range(NF):
        assert abs(float(np.mean(fl[0]["mpjpe"][step])) - g["mpjpe"][step]) < 1e-3 * g["mpjpe"][step]
        assert abs(float(np.mean(fl[0]["pampjpe"][step])) - g["pampjpe"][step]) < 2e-3 * g["pampjpe"][step]
        assert abs(float(np.ravel(fl[0]["pve"])[step]) - g["pve"][step]) < 1e-3 * g["pve"][step]
    # (the throughput schedule sums in another order than the reference AND than a sequence alone: same class-pooled fp32 floor)
    assert_final_state_matches_golden(ads[0], g, theta0, {}, tag="fo_inner3_frameonly")
    # every replica against itself adapted alone: summation order differs (replica-aware split, chunked GroupNorm), arithmetic not
    for r in range(S):
        a = ads[r]
        st = a.optimizer.state[a.model.module.theta]
        assert st["step"] == NF
        # (Adam moves an element whose gradient is rounding noise by +-lr whatever its size, and the two schedules sum in different
        # orders: 1.08e-6 measured on one replica once the one-pass GroupNorm backward came in; the moments are the real check)
        assert _rel(a.model.module.theta.detach(), singles[r][0]) < 3e-6, r
        assert _rel(st["exp_avg"], singles[r][1]) < 5e-3, r
        assert _rel(st["exp_avg_sq"], singles[r][2]) < 1e-2, r
        for k in ("mpjpe", "pampjpe", "pve"):
            np.testing.assert_allclose(np.ravel(np.array(fl[r][k], np.float64)), np.ravel(np.array(singles[r][3][k], np.float64)), rtol=2e-3)
    assert not torch.equal(ads[0].model.module.theta.detach(), ads[1].model.module.theta.detach())


def test_headline_32_sequences_one_frame(headline_switches):
    """S = 32 (bench.py's default sequences per GPU), one frame: replica 0 against the reference's first frame, every replica
    against its own single-sequence run."""
    lib = headline_switches
    S, NF = 32, 1
    assert _throughput_schedule_is_on(lib, S)
    from dynaboa_amd import native_step as NS
    g = golden("g5_fo_inner3_frameonly.npz")
    frames = _frames(S, NF)
    singles = _singles(S, NF, frames, lib)
    ads = [_mk(r) for r in range(S)]
    grp = NS.ReplicaGroup(ads, NF)
    grp.step([frames[r][0] for r in range(S)], 0)
    up = float(grp.stepper.losses(0, 3, 0)[3])
    assert abs(up - g["upper_loss"][0]) < 1e-4 * abs(g["upper_loss"][0])
    a0 = ads[0]
    a0.model.eval()
    with torch.no_grad():
        r_, s_, c_ = a0.model(frames[0][0]["image"])
        j_ = a0.decode_smpl_params(r_, s_)["s3d"]
    for k, v in dict(rotmat=r_, shape=s_, cam=c_, joints=j_).items():
        assert rel_err(v.cpu().numpy(), g[f"pred0_{k}"]) < 1e-3, k
    fl = grp.flush_metrics()
    assert abs(float(np.mean(fl[0]["mpjpe"][0])) - g["mpjpe"][0]) < 1e-3 * g["mpjpe"][0]
    assert_first_frame_outer_gradient(ads[0], g)          # the throughput schedule's outer gradient against the reference's, tensor by tensor
    for r in range(S):
        a = ads[r]
        st = a.optimizer.state[a.model.module.theta]
        # after ONE Adam step every element has moved by lr * g / (|g| + eps) = +-3e-6, whatever its size: elements whose gradient is
        # rounding noise flip sign between summation orders (measured 1.3e-6 relative on theta = ~0.5 % of the elements; the 4-frame
        # test above holds 1e-6).  The first moment m = 0.1 g is linear in the gradient and is the real check.
        assert _rel(a.model.module.theta.detach(), singles[r][0]) < 5e-6, r
        assert _rel(st["exp_avg"], singles[r][1]) < 5e-3, r
        np.testing.assert_allclose(np.ravel(np.array(fl[r]["mpjpe"], np.float64)), np.ravel(np.array(singles[r][3]["mpjpe"], np.float64)), rtol=2e-3)


def test_ranged_weight_updates_beside_the_forward_change_nothing(headline_switches, monkeypatch):
    """Replica groups update the weights by arena ranges - [stem .. layer2] on the chain's stream, [layer3] and [layer4 + regressor] on
    the auxiliary stream beside the next forward's first layers (adapt_step.hip weight_update) - and the forward waits for each range
    before its first reader: two frames of S = 8 sequences with the ranged updates on (default) and off give the same weights and Adam
    state bit for bit (same kernels, same per-element arithmetic; only the streams differ)."""
    from dynaboa_amd import native_step as NS
    S, NF = 8, 2
    frames = _frames(S, NF)
    outs = []
    for ovl in ("1", "0"):
        monkeypatch.setenv("DYB_UPD_OVERLAP", ovl)
        ads = [_mk(r) for r in range(S)]
        grp = NS.ReplicaGroup(ads, NF)
        for step in range(NF):
            grp.step([frames[r][step] for r in range(S)], step)
        fl = grp.flush_metrics()
        row = []
        for r in range(S):
            st = ads[r].optimizer.state[ads[r].model.module.theta]
            row += [ads[r].model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                    torch.from_numpy(np.ravel(np.array(fl[r]["mpjpe"], np.float64)))]
        outs.append(row)
        del grp, ads
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_bench_multi_rank_control_flow_on_one_gpu():
    """`bench.py --gpus 2` end to end on this ONE-GPU box (DYB_BENCH_SMOKE_ONE_GPU=1: both ranks on cuda:0, gloo instead of
    RCCL, which refuses two ranks on one device) - the self-spawn through torch.distributed.run, rank / world from the environment,
    barrier + max-over-ranks timing, the ragged end-of-run gather of the per-frame errors (sharding.gather_frame_metrics) and the
    3DPW operating point at ceil(37 / N) sequences per GPU.  A functional check of the N > 1 path the driver runs on an 8-GPU node,
    never a measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DYB_BENCH_SMOKE_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    seqs, steps = 3, 2
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--seqs", str(seqs),
                          "--no_cpu_baseline", "--no_roofline", "--percentile_frames", "0"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                       # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == steps and rec["scaling"] == "weak"
    cfg = rec["config"]
    assert cfg["gathered_frames"] == steps * seqs * 2, cfg
    assert cfg["gathered_frames_per_rank"] == [steps * seqs, steps * seqs], cfg
    assert abs(rec["value"] - steps * seqs * 2 / (rec["ms_per_step"] * 1e-3 * steps)) < 1e-6 * rec["value"]
    op = rec["pw3d_operating_point"]
    assert op.get("value") and op["sequences_per_gpu"] == 19, op     # ceil(37 / 2)


def test_fast_weights_from_the_weight_gradient_epilogue_are_bit_identical_to_the_streaming_pass(headline_switches, monkeypatch):
    """"fuse_fast" (round 6): on the throughput schedule a lower level's unsplit weight gradients write theta_next = theta_cur - fastlr * g
    from their accumulators (the levels ping-pong between two fast-weight buffers; the streaming pass covers the rest of the arena by
    segments) - against the round-5 form (every gradient to HBM, one streaming pass over the whole arena): S = 8, three frames of
    3 inner + 1 outer step - weights, Adam moments and metrics bit for bit.  The same for "fuse_adam": the OUTER level's unsplit weight
    gradients apply Adam to theta / exp_avg / exp_avg_sq in place from their accumulators (dyb_adam_one, shared with the streaming kernel),
    the streaming Adam pass covers the remaining segments - all four combinations of the two switches must agree bit for bit."""
    from dynaboa_amd import native_step as NS
    S, NF = 8, 3
    frames = _frames(S, NF)
    outs = []
    for fuse, adam in (("0", "0"), ("1", "0"), ("1", "1"), ("0", "1")):
        monkeypatch.setenv("DYB_FUSE_FAST", fuse)           # read when the stepper is created
        monkeypatch.setenv("DYB_FUSE_ADAM", adam)           # the outer level's weight gradients apply Adam themselves
        # (the regressor's fc1 / fc2 / decoder matrices - 13 % of the parameters - take the same updates in linear_outer_kernel's epilogue
        # whenever a scope is open: covered by the same comparison against the ("0", "0") run)
        ads = [_mk(r) for r in range(S)]
        grp = NS.ReplicaGroup(ads, NF)
        for step in range(NF):
            grp.step([frames[r][step] for r in range(S)], step)
        fl = grp.flush_metrics()
        st = [a.optimizer.state[a.model.module.theta] for a in ads]
        outs.append([torch.stack([a.model.module.theta.detach() for a in ads]), torch.stack([s_["exp_avg"] for s_ in st]),
                     torch.stack([s_["exp_avg_sq"] for s_ in st]),
                     torch.tensor(np.array([np.ravel(np.array(fl[r]["mpjpe"], np.float64)) for r in range(S)]))])
        del grp, ads
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b), float((a.double() - b.double()).norm() / b.double().norm())


def test_fast_weights_from_the_weight_gradient_epilogue_one_sequence(monkeypatch):
    """"fuse_fast" on the latency schedule (ONE sequence, the literal bs=1 stream): every lower-level weight gradient - unsplit, folded
    in-kernel or by the fold launch - and the regressor's matrices write the fast weights themselves; against the streaming pass over the
    whole arena (DYB_FUSE_FAST=0, DYB_FUSE_LINEAR=0): three frames of 3 inner + 1 outer step, weights / Adam moments / metrics bit for bit."""
    frames = _frames(1, 3)[0]
    outs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("DYB_FUSE_FAST", fuse)
        ad = _mk(0)
        res = ad.excute(frames, nframes=3)
        assert ad._native is not None
        st = ad.optimizer.state[ad.model.module.theta]
        outs.append([ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                     torch.tensor(np.ravel(np.array(res["mpjpe"], np.float64)))])
        del ad
    for a, b in zip(*outs):
        assert torch.equal(a, b), float((a.double() - b.double()).norm() / b.double().norm())
