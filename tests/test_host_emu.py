"""Host-layer logic (module surface, arena packing, MAML / Adam plumbing, the per-frame schedule)
exercised on CPU by pointing dynaboa_amd at the emulator build of the kernels (tests/emu).
The full-frame case is opt-in (DYB_EMU_FULL=1, ~4 min); the GPU suite covers it at speed."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import cosine, golden, rel_err


@pytest.fixture(scope="module")
def emu_lib():
    from emu.build_emu import build
    from dynaboa_amd import _abi, _lib
    lib = _abi.bind(ctypes.CDLL(build()))
    saved = _lib._lib
    _lib.use_library(lib)
    yield lib
    _lib._lib = saved


def test_abi_header_matches_library(emu_lib):
    from dynaboa_amd import _abi
    protos = _abi.parse_header()
    assert len(protos) >= 45
    for name in protos:
        assert hasattr(emu_lib, name), name


def test_public_header_is_plain_c(tmp_path):
    """include/dynaboa_hip.h is the contract other host languages bind (cgo / JNI / ctypes): it must compile as
    C99 and as C++17 on its own - no torch, HIP or C++ types in any signature."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "dynaboa_hip.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(root, "include")
    if shutil.which("gcc"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    if shutil.which("g++"):
        subprocess.check_call(["g++", "-std=c++17", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
    assert shutil.which("gcc") or shutil.which("g++")


def test_plain_c_consumer_links_and_agrees_with_the_python_binding(emu_lib, tmp_path):
    """tests/c_abi/consumer.c - plan + stepper through nothing but include/dynaboa_hip.h - compiled by gcc, linked against the library
    under test, run: the sizes it reads are the ones the Python binding reads, option keys are checked by name."""
    import shutil
    import subprocess
    from emu.build_emu import build
    from dynaboa_amd.hmr import get_layout
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = build()
    exe = str(tmp_path / "consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "consumer.c"), "-L", os.path.dirname(lib), "-ldynaboa_emu",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr)
    got = {k: int(v) for k, v in (line.split() for line in out.stdout.strip().splitlines())}
    L = get_layout(1)
    assert got["param_floats"] == L.n_params and got["act_floats"] == L.act_floats and got["workspace_bytes"] == L.ws_bytes
    assert got["tensors"] == len(L.tensors)
    assert got["record_floats"] >= 2 * 14 * 3 + 1 + 1 and got["loss_floats"] == (3 + 1) * 4       # (records are padded to a float4 multiple)
    assert got["hvp_dual_floats"] > L.act_floats and got["stepper_workspace_bytes"] > 4 * got["workspace_bytes"]


def test_arena_pack_unpack_roundtrip(emu_lib, ckpt_rand):
    from dynaboa_amd.hmr_layout import HmrLayout
    L = HmrLayout(emu_lib, 1)
    assert L.n_params % 4 == 0
    flat = L.pack(ckpt_rand)
    back = L.unpack(flat)
    n = 0
    for k, v in ckpt_rand.items():
        if k.startswith("init_"):
            continue
        assert torch.equal(back[k], v), k
        n += v.numel()
    assert n == 26_977_501
    assert float(flat.abs().sum()) == pytest.approx(sum(float(v.abs().sum()) for k, v in ckpt_rand.items()
                                                        if not k.startswith("init_")), rel=1e-5)


def test_maml_second_order_matches_autograd(emu_lib):
    """Second-order adapt(): K inner steps through _SecondOrderStep (finite-difference Hessian-vector products over
    first-order gradients, with the loss closure) against plain torch create_graph=True on a smooth toy loss."""
    from dynaboa_amd.maml import MAML

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(5)
            self.theta = torch.nn.Parameter(torch.randn(256, generator=g) * 0.5)

    g = torch.Generator().manual_seed(6)
    A = torch.randn(256, 256, generator=g) / 16
    c = torch.randn(256, generator=g)

    def lower(th):
        return (torch.sin(th @ A) * c).sum() + 0.1 * (th ** 4).sum()

    def upper(th):
        return ((th - 0.3) ** 2).sum() + torch.cos(th).sum()

    toy = Toy()
    lr = 0.05
    # reference: explicit functional unroll with create_graph (what learn2learn first_order=False does)
    th = toy.theta
    fast = th
    for _ in range(3):
        (gk,) = torch.autograd.grad(lower(fast), [fast], create_graph=True)
        fast = fast - lr * gk
    (g_ref,) = torch.autograd.grad(upper(fast), [th])
    fast_fo = th.detach().clone().requires_grad_(True)
    cur = fast_fo
    for _ in range(3):
        (gk,) = torch.autograd.grad(lower(cur), [cur])
        cur = cur - lr * gk.detach()
    (g_fo,) = torch.autograd.grad(upper(cur), [fast_fo])

    old = MAML.fd_rel
    MAML.fd_rel = 1e-3           # toy curvature is O(1): a larger relative step keeps fp32 rounding out of the difference
    try:
        maml = MAML(toy, lr=lr, first_order=False)
        learner = maml.clone()
        for _ in range(3):
            learner.adapt(lower(learner._theta), closure=lambda l: lower(l._theta))
        toy.theta.grad = None
        upper(learner._theta).backward()
    finally:
        MAML.fd_rel = old
    e_so = rel_err(toy.theta.grad.numpy(), g_ref.numpy())
    gap = rel_err(g_fo.numpy(), g_ref.numpy())
    assert gap > 0.05, gap                    # the toy must separate first from second order
    assert e_so < 2e-3 and e_so < 0.05 * gap, (e_so, gap)
    l2 = maml.clone()
    with pytest.raises(NotImplementedError):
        l2.adapt(lower(l2._theta))            # no closure in second-order mode


def test_maml_adam_plumbing_small(emu_lib):
    """clone()/adapt() first-order semantics and the fused Adam on a toy 'module' with a flat theta."""
    from dynaboa_amd.maml import MAML, _FastWeightStep
    from dynaboa_amd.optim import Adam, ema_update

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.theta = torch.nn.Parameter(torch.linspace(-1, 1, 64))

        def forward(self, x, theta=None):
            t = self.theta if theta is None else theta
            return (t * x).sum()

    toy = Toy()
    m = MAML(toy, lr=0.1, first_order=True)
    x = torch.arange(64.0) / 64
    learner = m.clone()
    inner = (learner(x) - 1.0) ** 2
    learner.adapt(inner)
    g_inner = 2 * ((toy.theta.detach() * x).sum() - 1.0) * x
    assert torch.allclose(learner._theta.detach(), toy.theta.detach() - 0.1 * g_inner, atol=1e-6)
    outer = (learner(x) ** 2)
    opt = Adam(m.parameters(), lr=3e-3, betas=(0.5, 0.9))
    opt.zero_grad()
    outer.backward()
    fast = learner._theta.detach()
    assert torch.allclose(toy.theta.grad, 2 * (fast * x).sum() * x, atol=1e-5)       # FO: grad taken AT the fast weights
    ref = toy.theta.detach().clone().requires_grad_(True)
    ropt = torch.optim.Adam([ref], lr=3e-3, betas=(0.5, 0.9))
    ref.grad = toy.theta.grad.clone()
    ropt.step()
    opt.step()
    assert torch.allclose(toy.theta.detach(), ref.detach(), atol=1e-7)
    so = MAML(toy, 0.1, first_order=False).clone()
    with pytest.raises(NotImplementedError):
        so.adapt((so(x) - 1.0) ** 2)            # second order needs the loss closure (maml.py)
    t = torch.zeros(64)
    ema_update([t], [toy.theta.detach()], 0.1)
    assert torch.allclose(t, 0.9 * toy.theta.detach(), atol=1e-7)


def test_smpl_wrapper_surface(emu_lib, smpl_tabs):
    from dynaboa_amd.smpl import SMPL
    from oracle import ref_cpu as O
    smpl = SMPL(tables=smpl_tabs)
    g = torch.Generator().manual_seed(0)
    betas = torch.randn(1, 10, generator=g) * 0.5
    pose = torch.randn(1, 72, generator=g) * 0.2
    out = smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3])            # pose2rot=True path
    T = O.smpl_tables_to_torch(smpl_tabs)
    v, j = O.smpl_forward(T, betas, pose[:, 3:], pose[:, :3], pose2rot=True)
    assert out.vertices.shape == (1, 6890, 3) and out.joints.shape == (1, 49, 3)
    assert rel_err(out.vertices.numpy(), v.numpy()) < 1e-5 and rel_err(out.joints.numpy(), j.numpy()) < 1e-5


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~4 min under the emulator; set DYB_EMU_FULL=1")
def test_full_frame_on_emulator_matches_reference(emu_lib):
    from dynaboa_amd import assets
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    g = golden("g5_fo_inner1_frameonly_identity.npz")
    o = DB.frame_only_options(inner_step=1)
    ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True, randomize_norm=True), device="cpu")
    ad.reset_records(1)
    ad.global_step = 0
    batch = assets.make_frame(0, 1, seed=22)
    ad.model.eval()
    theta0 = ad.model.module.theta.detach().clone()
    mpjpe, pampjpe, pve = ad.adaptation(batch)
    up = float(ad.fit_losses["ul/total"])
    assert abs(up - g["upper_loss"][0]) < 1e-4 * abs(g["upper_loss"][0])
    assert abs(float(np.mean(mpjpe)) - g["mpjpe"][0]) < 1e-3 * g["mpjpe"][0]
    assert abs(float(np.mean(pampjpe)) - g["pampjpe"][0]) < 2e-3 * g["pampjpe"][0]
    with torch.no_grad():
        r, s, c = ad.model(batch["image"])
    assert rel_err(r.numpy(), g["pred0_rotmat"]) < 1e-3 and rel_err(c.numpy(), g["pred0_cam"]) < 1e-3
    assert float((ad.model.module.theta.detach() - theta0).abs().max()) > 0


@pytest.mark.slow
def test_native_stepper_is_bit_identical_to_autograd_path(emu_lib):
    """csrc/adapt_step.hip (one C call per frame) against the torch.autograd composition of the same kernels on the
    emulator: weights and Adam moments must be IDENTICAL, metric records / logged losses equal to rounding."""
    from dynaboa_amd import assets
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    frames = [assets.make_frame(0, 1, seed=22)]
    outs = []
    for native in (1, 0):
        o = DB.frame_only_options(inner_step=1)
        o.native_step, o.deferred_metrics = native, 1
        ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True, randomize_norm=True), device="cpu")
        res = ad.excute(frames, nframes=1)
        assert (ad._native is not None) == bool(native)
        st = ad.optimizer.state[ad.model.module.theta]
        outs.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"], res,
                     ad.metric_records, {k: float(v) for k, v in ad.last_summaries.items()}))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3] == 1
    for k in ("mpjpe", "pampjpe", "pve"):
        np.testing.assert_allclose(np.ravel(np.array(a[4][k], np.float64)), np.ravel(np.array(b[4][k], np.float64)), rtol=1e-5)
    assert [(r["step"], r["tag"]) for r in a[5]] == [(r["step"], r["tag"]) for r in b[5]] == [(0, ('lower', 0)), (0, ('final', 0))]
    for x, y in zip(a[5], b[5]):
        np.testing.assert_allclose(np.ravel(x["mpjpe"]), np.ravel(y["mpjpe"]), rtol=1e-5)
        np.testing.assert_allclose(np.ravel(x["pampjpe"]), np.ravel(y["pampjpe"]), rtol=1e-5)
    assert a[6].keys() == b[6].keys()
    for k in a[6]:
        assert a[6][k] == b[6][k], k
    g = golden("g5_fo_inner1_frameonly_identity.npz")
    assert abs(a[6]["ul/total"] - g["upper_loss"][0]) < 1e-4 * abs(g["upper_loss"][0])


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~3 min under the emulator; set DYB_EMU_FULL=1")
def test_native_stepper_side_stream_work_from_the_helper_thread(emu_lib, monkeypatch):
    """"side_thread": the side stream's launches (previous frame's final forward + record, this frame's ground-truth meshes) issued by the
    library's helper thread while the calling thread goes on with the chain.  The emulator executes a launch where it is issued, so the
    two threads really interleave here: weights, Adam state and every metric record must equal the in-line run bit for bit - i.e. the
    calling thread waits for the helper before it touches anything the helper writes."""
    from types import SimpleNamespace
    from dynaboa_amd import assets, benchmark as DB, native_step as NS
    from dynaboa_amd._abi import check
    from dynaboa_amd.base_adaptor import synthetic_bundle
    frames = [assets.make_frame(s, 1, seed=22) for s in range(3)]
    orig = NS.NativeStepper.adapt_frames
    outs = []
    for helper in (0, 1):
        def wrapped(self, batches, side_stream=None, helper=helper):
            self.use_side = 1
            check(self.lib.dyb_stepper_set_i(self.h, b"use_side", 1), "use_side")
            check(self.lib.dyb_stepper_set_i(self.h, b"side_thread", helper), "side_thread")
            return orig(self, batches, SimpleNamespace(cuda_stream=2))       # any non-null handle is a stream to the emulator
        monkeypatch.setattr(NS.NativeStepper, "adapt_frames", wrapped)
        o = DB.frame_only_options(inner_step=1)
        o.native_step, o.deferred_metrics = 1, 1
        ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True, randomize_norm=True), device="cpu")
        res = ad.excute(frames, nframes=3)
        assert ad._native is not None
        st = ad.optimizer.state[ad.model.module.theta]
        outs.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                     np.ravel(np.array(res["pampjpe"], np.float64)), np.ravel(np.array(res["mpjpe"], np.float64)),
                     np.ravel(np.array(res["pve"], np.float64))))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0][3:], outs[1][3:]):
        assert a.size == 3 and np.isfinite(a).all()                          # one final record per frame
        np.testing.assert_array_equal(a, b)


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~4 min under the emulator; set DYB_EMU_FULL=1")
def test_native_stepper_side_stream_schedule_under_adversarial_stream_order(emu_lib, monkeypatch):
    """The native frame step with its weight-gradient convolutions on the auxiliary stream, in the emulator's lazy stream mode (see
    kernel_cases.case_stream_order): two frames drained chain-first and side-stream-first give the weights / Adam state / records of
    the in-line run bit for bit - every consumer of a side-stream result (fast-weight update, Adam, the next frame) waits for it."""
    from types import SimpleNamespace
    from dynaboa_amd import _lib, assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    raw = _lib.load()
    frames = [assets.make_frame(s, 1, seed=22) for s in range(2)]
    orig = NS.NativeStepper.adapt_frames
    outs = []
    for order in (None, 0, 1):
        used = []

        def wrapped(self, batches, side_stream=None, order=order, used=used):
            if order is None:
                return orig(self, batches, side_stream)
            self._aux = SimpleNamespace(cuda_stream=1)          # any non-null handle is a second stream to the emulator
            raw.emu_lazy(1)
            try:
                return orig(self, batches, side_stream)
            finally:
                used.append(raw.emu_flush(order))
                raw.emu_lazy(0)
        monkeypatch.setattr(NS.NativeStepper, "adapt_frames", wrapped)
        o = DB.frame_only_options(inner_step=1)
        o.native_step, o.deferred_metrics = 1, 1
        ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True, randomize_norm=True), device="cpu")
        res = ad.excute(frames, nframes=2)
        assert ad._native is not None
        if order is not None:
            assert used == [2, 2], used                          # both queues held work in both frame steps
        st = ad.optimizer.state[ad.model.module.theta]
        outs.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                     np.ravel(np.array(res["pampjpe"], np.float64)), np.ravel(np.array(res["mpjpe"], np.float64))))
    for other in outs[1:]:
        for a, b in zip(outs[0][:3], other[:3]):
            assert torch.equal(a, b)
        np.testing.assert_array_equal(outs[0][3], other[3])
        np.testing.assert_array_equal(outs[0][4], other[4])


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~6 min under the emulator; set DYB_EMU_FULL=1")
@pytest.mark.parametrize("throughput", [0, 1], ids=["latency_schedule", "throughput_schedule"])
def test_replica_group_ranged_weight_updates_under_adversarial_stream_order(emu_lib, monkeypatch, throughput):
    """Replica groups update the weights by arena ranges: [stem .. layer2] on the chain's stream, [layer3] and [layer4 + regressor] on
    the auxiliary stream beside the next forward's first layers, which waits for each range right before its first reader
    (adapt_step.hip weight_update / DybFwdGates).  One frame step of S = 2 sequences (a fast-weight step, Adam, the final inference)
    in the emulator's lazy stream mode, drained chain-first and auxiliary-stream-first: weights / Adam state / metrics of the in-line
    run bit for bit.  Round 6: the fast weights / Adam's step of most tensors are written by the weight gradients' own epilogues ON THE
    AUXILIARY STREAM (fuse_fast / fuse_adam) - on the latency schedule (in-kernel fold / fold launch forms) and, `throughput`, on the
    throughput schedule (rep_split 1, tp_min 1: the unsplit igemm_tp epilogues incl. Adam in place on theta) - so the readers of those
    weights (the next forward, the final inference) must be ordered behind them by the backward's join, not by the update's events."""
    from types import SimpleNamespace
    from dynaboa_amd import _lib, assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    raw = _lib.load()
    if throughput:
        raw.dyb_set_option(b"rep_split", 1)
        raw.dyb_set_option(b"tp_min", 1)
        monkeypatch.setattr(NS, "set_replica_policy", lambda *a, **k: None, raising=False)
    S = 2
    frames = [assets.make_frame(100 * r, 1, seed=22) for r in range(S)]
    orig = NS.NativeStepper.adapt_frames
    outs = []
    for order in (None, 0, 1):
        used = []

        def wrapped(self, batches, side_stream=None, order=order, used=used):
            self._aux = SimpleNamespace(cuda_stream=1)          # any non-null handle is a second stream to the emulator
            if order is None:
                return orig(self, batches, side_stream)
            raw.emu_lazy(1)
            try:
                return orig(self, batches, side_stream)
            finally:
                used.append(raw.emu_flush(order))
                raw.emu_lazy(0)
        monkeypatch.setattr(NS.NativeStepper, "adapt_frames", wrapped)
        ads = []
        for r in range(S):
            o = DB.frame_only_options(inner_step=1)
            o.deferred_metrics = 1
            ads.append(DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True), device="cpu"))
        grp = NS.ReplicaGroup(ads, 1)
        grp.step(frames, 0)
        fl = grp.flush_metrics()
        if order is not None:
            assert used == [2], used
        row = []
        for r in range(S):
            st = ads[r].optimizer.state[ads[r].model.module.theta]
            row += [ads[r].model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                    torch.from_numpy(np.ravel(np.array(fl[r]["mpjpe"], np.float64)))]
        outs.append(row)
    if throughput:
        raw.dyb_set_option(b"rep_split", 0)
        raw.dyb_set_option(b"tp_min", 8)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~10 min under the emulator; set DYB_EMU_FULL=1")
def test_full_term_replica_group_ranged_updates_under_adversarial_stream_order(emu_lib, monkeypatch):
    """The reference's default term set for S = 2 replicas with the ranged weight updates (fast-weight step, Adam + the teacher's EMA by
    arena ranges on the auxiliary stream) in the emulator's lazy stream mode, drained chain-first and auxiliary-stream-first: weights,
    Adam state and teacher of the in-line run bit for bit."""
    from types import SimpleNamespace
    from dynaboa_amd import _lib, assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    raw = _lib.load()
    S = 2
    frames = [assets.make_frame(100 * r, 1, seed=22) for r in range(S)]
    orig = NS.NativeStepper.adapt_frames_full
    outs = []
    for order in (None, 0, 1):
        used = []

        def wrapped(self, *a, order=order, used=used, **kw):
            self._aux = SimpleNamespace(cuda_stream=1)          # any non-null handle is a second stream to the emulator
            if order is None:
                return orig(self, *a, **kw)
            raw.emu_lazy(1)
            try:
                return orig(self, *a, **kw)
            finally:
                used.append(raw.emu_flush(order))
                raw.emu_lazy(0)
        monkeypatch.setattr(NS.NativeStepper, "adapt_frames_full", wrapped)
        ads = []
        for r in range(S):
            o = DB.parser.parse_args([])
            o.inner_step, o.interval, o.dynamic_boa, o.deferred_metrics = 1, 2, 0, 1     # (no dynamic-BOA gate: its host poll cannot run inside a lazy section)
            ads.append(DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True), device="cpu"))
        grp = NS.ReplicaGroup(ads, 1)
        assert grp.stepper.full
        grp.step(frames, 0)
        grp.flush_metrics()
        if order is not None:
            assert used and used[0] >= 2, used                   # chain + auxiliary stream (+ the pass streams of a small group, round 5)
        row = []
        for r in range(S):
            st = ads[r].optimizer.state[ads[r].model.module.theta]
            row += [ads[r].model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), ads[r].teacher.theta.detach().clone()]
        outs.append(row)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~8 min under the emulator; set DYB_EMU_FULL=1")
def test_full_term_parallel_passes_under_adversarial_stream_order(emu_lib, monkeypatch):
    """One sequence, the reference's default term set: the history pass and the exemplar pass of a level run on two streams of the
    stepper's own beside the frame pass (adapt_step.hip full_level, "par_passes").  Three frames (the third has a history frame:
    interval 1) in the emulator's lazy stream mode, drained chain-first and in reverse: weights, Adam state, teacher and metrics of the
    sequential run (par_passes = 0) bit for bit - every cross-stream read waits for its event."""
    from types import SimpleNamespace
    from dynaboa_amd import _lib, assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    raw = _lib.load()
    frames = [assets.make_frame(s, 1, seed=22) for s in range(3)]
    orig = NS.NativeStepper.adapt_frame_full
    outs = []
    for par, order in ((0, None), (1, None), (1, 0), (1, 1)):
        used = []

        def wrapped(self, *a, order=order, used=used, par=par, **kw):
            self._aux = SimpleNamespace(cuda_stream=1)          # any non-null handle is a second stream to the emulator
            assert self.lib.dyb_stepper_set_i(self.h, b"par_passes", par) == 0
            if order is None:
                return orig(self, *a, **kw)
            raw.emu_lazy(1)
            try:
                return orig(self, *a, **kw)
            finally:
                used.append(raw.emu_flush(order))
                raw.emu_lazy(0)
        monkeypatch.setattr(NS.NativeStepper, "adapt_frame_full", wrapped)
        o = DB.parser.parse_args([])
        o.inner_step, o.interval, o.dynamic_boa, o.deferred_metrics = 1, 1, 0, 1     # (no dynamic-BOA gate: its host poll cannot run inside a lazy section)
        ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=False, randomize_norm=True), device="cpu")
        res = ad.excute(frames, nframes=3)
        assert ad._native is not None and ad._native.full
        if order is not None:
            assert used[0] >= 3 and used[-1] == 4, used          # chain + auxiliary + exemplar stream; + the history stream once there is one
        st = ad.optimizer.state[ad.model.module.theta]
        outs.append([ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), ad.teacher.theta.detach().clone(),
                     torch.from_numpy(np.ravel(np.array(res["mpjpe"], np.float64)))])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_replica_policy_switches(emu_lib):
    """native_step.set_replica_policy (what bench.py and the sharded driver call for several sequences per GPU): the replica-aware
    split policy and the throughput schedule from TP_MIN_SEQUENCES sequences per launch; 'bitexact' leaves the single-sequence policy."""
    from dynaboa_amd import benchmark as DB, native_step as NS

    def opt(name):
        v = ctypes.c_int(-1)
        assert emu_lib.dyb_get_option(name, ctypes.byref(v)) == 0
        return v.value
    try:
        NS.set_replica_policy(True)
        assert opt(b"rep_split") == 1 and opt(b"tp_min") == NS.TP_MIN_SEQUENCES == 5
        NS.set_replica_policy(False)
        assert opt(b"rep_split") == 0
    finally:
        emu_lib.dyb_set_option(b"rep_split", 0)
        emu_lib.dyb_set_option(b"tp_min", 8)
    assert DB.parser.parse_args([]).replica_policy == "throughput"
    assert DB.parser.parse_args(["--replica_policy", "bitexact"]).replica_policy == "bitexact"
    assert opt(b"tp_batch_min") == 8 and opt(b"lat_fold") == 1 and opt(b"tp_fold") == 0 and opt(b"tp_wt") == 1      # round-5 defaults


def test_native_stepper_coverage_rules(emu_lib):
    from dynaboa_amd import benchmark as DB, native_step as NS
    assert NS.mode(DB.frame_only_options(inner_step=3)) == "frame" and NS.supported(DB.frame_only_options(inner_step=3)) is None
    assert NS.mode(DB.parser.parse_args([])) == "full"                            # the reference's default term set: covered too
    assert NS.supported(DB.frame_only_options(second_order=1)) == "second order"
    assert NS.supported(DB.frame_only_options(share_forwards=0)) is not None
    o = DB.parser.parse_args([])
    o.batch_size = 32
    assert NS.mode(o) == "" and "batch" in NS.supported(o)
    o = DB.parser.parse_args([])
    o.teacher_dropout = 1
    assert NS.mode(o) == "full"                                                   # train-mode teacher: on the stepper since round 6
    o = DB.parser.parse_args([])
    o.sample_num = 2
    assert NS.mode(o) == "" and NS.mode(DB.parser.parse_args([]), have_bundle=False) == "full"


@pytest.mark.slow
def test_frame_level_exact_hvp_matches_oracle_second_derivative(emu_lib, gmm_t):
    """--hvp exact: H v of a whole frame-loss level (HMR -> SMPL -> 2-D loss + priors) from the tangent passes + the head's
    difference quotient, against torch differentiating the oracle's level loss twice - per tensor, next to the default
    finite-difference product of the same level."""
    from oracle import ref_cpu as O
    from dynaboa_amd import assets, benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    from dynaboa_amd.hmr import get_layout
    from conftest import cosine
    o = DB.frame_only_options(inner_step=1, second_order=1, hvp="exact")
    ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True, randomize_norm=True), device="cpu")
    ad.model.eval()
    batch = assets.make_frame(0, 1, seed=22)
    image, kp = batch["image"], batch["smpl_j2d"]
    L = get_layout(1)
    theta = ad.model.module.theta.detach()
    # reference level loss on the oracle
    mp = assets.make_smpl_mean_params(identity_pose=True, seed=3)
    sd = assets.make_synthetic_checkpoint(22, mp, randomize_norm=True, prefix="")["model"]
    oa = O.Adapter(sd, O.smpl_tables_to_torch(assets.make_synthetic_smpl(0)), gmm_t,
                   dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0, use_motion=0, dynamic_boa=0,
                        use_temporal_losses_upper=0, inner_step=1))
    names = list(oa.theta)
    plist = [oa.theta[k] for k in names]
    rng = np.random.default_rng(9)
    vdict = {k: torch.from_numpy(rng.standard_normal(tuple(oa.theta[k].shape)).astype(np.float32)) * (0.02 if oa.theta[k].dim() > 1 else 0.05)
             for k in names}

    def level(w):
        rot, shape, cam = O.hmr_forward(oa._full(w), image)
        j49, _ = oa.decode(rot, shape)
        return oa.frame_losses(rot, shape, O.projection(cam, j49), kp, "ll")
    g = torch.autograd.grad(level(oa.theta), plist, create_graph=True)
    gv = sum((a * vdict[k]).sum() for a, k in zip(g, names))
    hv_ref = dict(zip(names, torch.autograd.grad(gv, plist)))
    # the same weights in the arena?  (the adaptor was built from the same seeds)
    P0 = L.unpack(theta)
    assert all(torch.equal(P0[k], oa.theta[k].detach()) for k in ("conv1.weight", "fc2.weight", "layer3.2.bn2.weight"))
    vfull = dict(vdict, **{k: torch.zeros_like(v) for k, v in oa.buf.items()})
    v = L.pack(vfull)
    learner = ad.model.clone()
    exact = ad.level_hvp_factory("lower", image, kp, learner)(theta)(v)
    H = L.unpack(exact)

    def errs(Hd):
        out = {}
        for k in names:
            b = hv_ref[k].double().flatten()
            if float(b.norm()) > 0:
                a = Hd[k].double().flatten()
                out[k] = (float((a - b).norm() / b.norm()), cosine(a.numpy(), b.numpy()))
        return out
    ee = errs(H)
    bad = {k: x for k, x in ee.items() if x[0] > 3e-3 or x[1] < 0.9999}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]
    # the default difference quotient of the same level, for the record (element-wise noisier in the backbone)
    ad.options.hvp = "fd"
    loss, _ = ad.lower_level_adaptation(image, kp, None, learner)
    closure = ad.level_closure("lower", image, kp, None)
    from dynaboa_amd.maml import MAML

    def grad_at(th):
        th = th.clone().requires_grad_(True)
        probe = MAML(ad.model.module, ad.model.lr, True, _theta=th)
        probe.eval()
        return torch.autograd.grad(closure(probe), [th])[0]
    eps = MAML.fd_rel * torch.linalg.vector_norm(theta) / torch.linalg.vector_norm(v)
    fd = (grad_at(theta + eps * v) - grad_at(theta - eps * v)) / (2 * eps)
    ef = errs(L.unpack(fd))
    print("exact: max rel %.2e  min cos %.7f | fd: max rel %.2e  min cos %.7f" % (
        max(x[0] for x in ee.values()), min(x[1] for x in ee.values()), max(x[0] for x in ef.values()), min(x[1] for x in ef.values())))


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~5 min under the emulator; set DYB_EMU_FULL=1")
def test_second_order_exact_hvp_first_frame_vs_reference_second_order(emu_lib):
    """--second_order 1 --hvp exact at the benchmarked depth (inner_step 3) on the emulator: the outer gradient of the first
    frame against the reference run with learn2learn first_order=False (golden g5_so_inner3_frameonly)."""
    from dynaboa_amd import assets
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    from conftest import cosine
    gso, gfo = golden("g5_so_inner3_frameonly.npz"), golden("g5_fo_inner3_frameonly.npz")
    o = DB.parser.parse_args([])
    for k, v in dict(retrieval=0, lower_level_mixtrain=0, upper_level_mixtrain=0, use_meanteacher=0, use_motion=0, dynamic_boa=0,
                     use_temporal_losses_upper=0, inner_step=3, second_order=1, hvp="exact").items():
        setattr(o, k, v)
    ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=False, randomize_norm=True, smpl_seed=0), device="cpu")
    ad.reset_records(1)
    ad.global_step = 0
    ad.model.eval()
    ad.adaptation(assets.make_frame(0, 1, seed=22))
    up = float(ad.fit_losses["ul/total"])
    assert abs(up - gso["upper_loss"][0]) < 1e-4 * abs(gso["upper_loss"][0])
    hmr = ad.model.module
    st = ad.optimizer.state[hmr.theta]
    g1 = hmr._layout1.unpack(st["exp_avg"] / (1 - ad.options.beta1))
    names = [str(x) for x in gso["names"]]
    gn = np.array([float(g1[k].double().norm()) for k in names])
    err = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
    gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
    sl = {k[3:]: (rel_err(g1[k[3:]].flatten()[:256].numpy(), gso[k]), cosine(g1[k[3:]].flatten()[:256].numpy(), gso[k]))
          for k in gso.files if k.startswith("g1_") and k not in ("g1_norms",)}
    print("exact SO inner3: grad-norm error median %.2e max %.2e (FO-SO gap median %.2e); slices %s" % (
        np.median(err), err.max(), np.median(gap), {k: ("%.1e" % a, "%.6f" % b) for k, (a, b) in sl.items()}))
    assert np.median(err) < 5e-4 and err.max() < 3e-3
    assert all(b > 0.9999 for _, b in sl.values()), sl


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~10 min under the emulator; set DYB_EMU_FULL=1")
def test_second_order_full_loss_set_exact_hvp_vs_reference_second_order(emu_lib):
    """--second_order 1 --hvp exact --hvp_terms all on the reference's default term set, first two frames on the emulator, against
    the reference run with learn2learn first_order=False (golden g5_so_inner1_full; the GPU test runs all four frames)."""
    from dynaboa_amd import assets
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    from conftest import cosine
    gso, gfo = golden("g5_so_inner1_full.npz"), golden("g5_fo_inner1_full.npz")
    o = DB.parser.parse_args([])
    for k, v in dict(inner_step=1, interval=2, optim_steps=2, second_order=1, hvp="exact", hvp_terms="all").items():
        setattr(o, k, v)
    ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=False, randomize_norm=True, smpl_seed=0), device="cpu")
    ad.reset_records(2)
    hmr = ad.model.module
    names = [str(x) for x in gso["names"]]
    for step in range(2):
        ad.global_step = step
        ad.fit_losses = {}
        batch = assets.make_frame(step, 1, seed=22)
        ad.model.eval()
        ad.adaptation(batch)
        up = float(ad.fit_losses["ul/total"])
        assert abs(up - gso["upper_loss"][step]) < 1e-4 * abs(gso["upper_loss"][step]), (step, up, gso["upper_loss"][step])
        with torch.no_grad():
            r, s, c = ad.model(batch["image"])
        for k, v in dict(rotmat=r, shape=s, cam=c).items():
            assert rel_err(v.numpy(), gso[f"pred{step}_{k}"]) < 1e-3, (step, k)
        if step == 0:
            st = ad.optimizer.state[hmr.theta]
            g1 = hmr._layout1.unpack(st["exp_avg"] / (1 - ad.options.beta1))
            gn = np.array([float(g1[k].double().norm()) for k in names])
            err = np.abs(gn - gso["g1_norms"]) / gso["g1_norms"]
            gap = np.abs(gfo["g1_norms"] - gso["g1_norms"]) / gso["g1_norms"]
            sl = {k[3:]: cosine(g1[k[3:]].flatten()[:256].numpy(), gso[k]) for k in gso.files if k.startswith("g1_") and k != "g1_norms"}
            print("SO full set exact: grad-norm error median %.2e max %.2e (FO-SO gap median %.2e max %.2e), min slice cosine %.6f" % (
                np.median(err), err.max(), np.median(gap), gap.max(), min(sl.values())))
            assert np.median(err) < 5e-4 and err.max() < 5e-3, (np.median(err), err.max())
            assert min(sl.values()) > 0.999, sl


def test_term_kernel_autograd_nodes_match_torch_composition(emu_lib):
    """losses.teacher_term / motion_term / labelled_term (dyb_aux_loss_terms as autograd nodes, what BaseAdaptor._level uses by default
    on the autograd path) against the torch composition of the reference's formulas (base_adaptor.py:320-343, :379-398, :346-376 +
    :412-422): value, logged components and the gradients autograd routes to both passes, with strided (state-slice) inputs."""
    import torch.nn.functional as F
    from dynaboa_amd import losses as LS
    from kernel_cases import _proj_t
    g = torch.Generator().manual_seed(17)
    rn = lambda *s: torch.randn(*s, generator=g)
    for B in (3, 20):                      # 20: beyond one launch's 16 samples (size-weighted chunks)
        _term_nodes_vs_torch(B, g, rn)


def _term_nodes_vs_torch(B, g, rn):
    import torch.nn.functional as F
    from dynaboa_amd import losses as LS
    from kernel_cases import _proj_t

    def pass_():
        rot = (torch.eye(3).expand(B, 24, 3, 3) + 0.3 * rn(B, 24, 3, 3)).clone().requires_grad_(True)
        state = torch.zeros(B, 160)
        state[:, 144:154] = rn(B, 10) * 0.5
        state[:, 154:157] = torch.tensor([0.9, 0.02, -0.03]) + 0.05 * rn(B, 3)
        state = state.requires_grad_(True)
        return rot, state, (rn(B, 49, 3) * 0.3).requires_grad_(True)
    kp = torch.cat([torch.rand(B, 49, 2, generator=g) * 2 - 1, (torch.rand(B, 49, 1, generator=g) < 0.7).float()], -1)
    kp2 = torch.cat([torch.rand(B, 49, 2, generator=g) * 2 - 1, (torch.rand(B, 49, 1, generator=g) < 0.7).float()], -1)
    gt_rot, gt_betas = torch.eye(3).expand(B, 24, 3, 3) + 0.3 * rn(B, 24, 3, 3), rn(B, 10) * 0.5
    gt_s3d = torch.cat([rn(B, 24, 3) * 0.3, torch.ones(B, 24, 1)], -1)
    for mode in (0, 1, 2):
        rot, state, joints = pass_()
        rot2, state2, joints2 = pass_()
        shape, cam, shape2, cam2 = state[:, 144:154], state[:, 154:157], state2[:, 144:154], state2[:, 154:157]
        s2d = _proj_t(cam, joints)
        if mode == 0:
            comps = [F.mse_loss(s2d, _proj_t(cam2, joints2).detach()), F.mse_loss(joints2.detach(), joints), F.mse_loss(shape, shape2.detach()),
                     F.mse_loss(rot, rot2.detach())]
            ref = comps[0] * 5 + comps[1] * 5 + comps[2] * 0.001 + comps[3]
            got, gc = LS.teacher_term(rot, shape, cam, joints, rot2.detach(), shape2.detach(), cam2.detach(), joints2.detach())
            leaves = [rot, state, joints]
        elif mode == 1:
            pm = s2d[:, 25:] - _proj_t(cam2, joints2)[:, 25:]
            gm = kp[:, 25:, :2] - kp2[:, 25:, :2]
            conf = ((kp2[:, 25:, 2:] + kp[:, 25:, 2:]) == 2).float()
            ref = (((pm - gm) ** 2) * conf).mean()
            comps = [ref]
            got, gc = LS.motion_term(rot, shape, cam, joints, cam2, joints2, kp, kp2)
            leaves = [state, joints, state2, joints2]
        else:
            conf = kp[:, 25:, 2:]
            p24, g24 = joints[:, 25:], gt_s3d[:, :, :3]
            pc, gcn = p24 - ((p24[:, 2] + p24[:, 3]) / 2)[:, None], g24 - ((g24[:, 2] + g24[:, 3]) / 2)[:, None]
            comps = [(((s2d[:, 25:] - kp[:, 25:, :2]) ** 2) * conf).mean(), (conf * (pc - gcn) ** 2).mean(), F.mse_loss(shape, gt_betas),
                     F.mse_loss(rot, gt_rot)]
            ref = comps[0] * 5 + comps[1] * 5 + comps[2] * 0.001 + comps[3]
            got, gc = LS.labelled_term(rot, shape, cam, joints, kp, gt_rot, gt_betas, gt_s3d)
            leaves = [rot, state, joints]
        assert abs(float(got) - float(ref)) < 1e-5 * abs(float(ref)), (mode, float(got), float(ref))
        for i, c in enumerate(comps):
            assert abs(float(gc[i]) - float(c)) < 1e-5 * abs(float(c)) + 1e-12, (mode, i)
        w = 0.37
        g_ref = torch.autograd.grad(ref * w, leaves)
        g_got = torch.autograd.grad(got * w, leaves)
        for a, b in zip(g_got, g_ref):
            assert rel_err(a.numpy(), b.numpy()) < 2e-5, (mode, rel_err(a.numpy(), b.numpy()))


def test_exact_hvp_selection_rules(emu_lib):
    """--hvp exact serves every level by default (--hvp_terms all); with --hvp_terms frame levels with teacher / motion / labelled
    terms (and any level with --hvp fd) get no factory, i.e. MAML.adapt differences the closure's gradient."""
    from dynaboa_amd import assets, benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    assert DB.parser.parse_args([]).hvp in ("exact", "fd")
    bundle = synthetic_bundle(seed=22, identity_pose=True)
    img, kp = assets.make_frame(0, 1, seed=22)["image"], assets.make_frame(0, 1, seed=22)["smpl_j2d"]
    ad = DB.Adaptor(DB.frame_only_options(inner_step=1, second_order=1, hvp="exact"), bundle, device="cpu")
    learner = ad.model.clone()
    assert ad.level_hvp_factory("lower", img, kp, learner) is not None
    ad.options.hvp = "fd"
    assert ad.level_hvp_factory("lower", img, kp, learner) is None
    full = DB.parser.parse_args([])                      # the reference's default term set: labelled exemplars in the lower level
    full.second_order, full.hvp = 1, "exact"
    assert full.hvp_terms == "all"                       # the multi-pass form is the default (checked on MI355X against the reference)
    ad2 = DB.Adaptor(full, bundle, device="cpu")
    assert ad2.level_hvp_factory("lower", img, kp, ad2.model.clone()) is not None
    ad2.options.hvp_terms = "frame"                      # exact products for frame-loss levels only
    assert ad2.level_hvp_factory("lower", img, kp, ad2.model.clone()) is None
    assert ad2.level_hvp_factory("upper", img, kp, ad2.model.clone()) is None


def test_fused_adam_accumulate_matches_separate_accumulate(emu_lib):
    """Second order: leaving the last accumulation (v - lr * H v) to dyb_adam_step_accum gives the Adam state of the two-launch
    sequence (fast-weight kernel, then dyb_adam_step) - toy loss, K = 2 inner steps."""
    from dynaboa_amd.maml import MAML
    from dynaboa_amd.optim import Adam, materialize_grad

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(5)
            self.theta = torch.nn.Parameter(torch.randn(256, generator=g) * 0.5)
    g = torch.Generator().manual_seed(6)
    A = torch.randn(256, 256, generator=g) / 16
    c = torch.randn(256, generator=g)
    lower = lambda th: (torch.sin(th @ A) * c).sum() + 0.1 * (th ** 4).sum()
    upper = lambda th: ((th - 0.3) ** 2).sum() + torch.cos(th).sum()
    old = MAML.fd_rel
    MAML.fd_rel = 1e-3
    res = []
    try:
        for defer in (False, True):
            toy = Toy()
            maml = MAML(toy, lr=0.05, first_order=False)
            maml.defer_accumulate = defer
            opt = Adam([toy.theta], lr=1e-2, betas=(0.5, 0.9))
            learner = maml.clone()
            for _ in range(2):
                learner.adapt(lower(learner._theta), closure=lambda l: lower(l._theta))
            opt.zero_grad()
            upper(learner._theta).backward()
            assert (getattr(toy.theta, "_so_pending", None) is not None) == defer
            full = materialize_grad(toy.theta).clone()
            opt.step()
            assert getattr(toy.theta, "_so_pending", None) is None
            st = opt.state[toy.theta]
            res.append((full, toy.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()))
    finally:
        MAML.fd_rel = old
    for a, b in zip(*res):
        assert rel_err(b.numpy(), a.numpy()) < 1e-6


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~6 min under the emulator; set DYB_EMU_FULL=1")
def test_second_order_full_loss_set_exact_hvp_vs_oracle(emu_lib, gmm_t, smpl_tabs):
    """Second order with the reference's default term set (labelled exemplars in the lower level; teacher + exemplars in the
    upper; frame 0, so no motion term) and exact Hessian-vector products of the multi-pass level (general_level_hvp): the
    outer gradient against the CPU oracle in create_graph=True mode, next to the oracle's first-order gradient."""
    from dynaboa_amd import assets
    from dynaboa_amd import benchmark as DB
    from dynaboa_amd.base_adaptor import synthetic_bundle
    from oracle import ref_cpu as O
    opts = dict(inner_step=1, interval=2, optim_steps=2, dynamic_boa=0)
    frame = assets.make_frame(0, 1, seed=22)
    o = DB.parser.parse_args([])
    for k, v in dict(opts, second_order=1, hvp="exact", hvp_terms="all").items():
        setattr(o, k, v)
    bundle = synthetic_bundle(seed=22, identity_pose=False, randomize_norm=True, smpl_seed=0)
    ad = DB.Adaptor(o, bundle, device="cpu")
    ad.reset_records(1)
    ad.global_step = 0
    ad.model.eval()
    ad.adaptation(frame)
    hmr_m = ad.model.module
    ours = hmr_m._layout1.unpack(ad.optimizer.state[hmr_m.theta]["exp_avg"] / (1 - ad.options.beta1))
    sd = {k.replace("module.", ""): v for k, v in bundle.checkpoint["model"].items()}
    grads = {}
    for so in (1, 0):
        ref = O.Adapter(sd, O.smpl_tables_to_torch(smpl_tabs), gmm_t, opts, first_order=not so)
        ref.exemplar_fn = lambda step: assets.make_exemplars(step, ref.o["sample_num"])
        grads[so] = ref.adapt_frame(frame)["outer_grad"]
    names = ["conv1.weight", "layer2.0.conv2.weight", "layer3.5.conv1.weight", "layer4.2.conv3.weight", "fc1.weight", "decpose.weight"]
    e_so = np.array([rel_err(ours[k].numpy(), grads[1][k].numpy()) for k in names])
    gap = np.array([rel_err(grads[0][k].numpy(), grads[1][k].numpy()) for k in names])
    print("SO full-loss exact: err vs oracle-SO", e_so, " FO-vs-SO gap", gap)
    # conv1: the FIRST-order gradient of this configuration already differs by 4e-2 (max-abs; 8e-3 in norm, cosine 0.99997)
    # between the engine and the oracle - a few stem activations within fp32 rounding of zero flip their ReLU mask - so the
    # stem row is bounded by that, the rest by the Hessian-vector products' own accuracy
    # (layer2.0.conv2 / layer3.5.conv1 sit at 4.3e-3 / 2.1e-3 on this checkpoint - the same mask flips further up the backbone;
    # identical at the round-4 tree and with every round-5 switch off - the head-side rows at 1e-6)
    assert e_so[0] < 6e-2 and (e_so[1:3] < 8e-3).all() and (e_so[3:] < 1e-3).all(), e_so
    assert (e_so[1:] < 0.03 * gap[1:] + 1e-5).all(), (e_so, gap)


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~4 min under the emulator; set DYB_EMU_FULL=1")
@pytest.mark.parametrize("throughput", [0, 1])
def test_replica_group_on_emulator(emu_lib, throughput):
    """S = 2 sequences in one launch grid (replicas: csrc/dyb_common.h) on the emulator against the same sequences adapted one
    at a time: identical weights / Adam state with the single-sequence policy; with the replica-aware policy forced into the
    throughput schedule (rep_split = 1, tp_min = 2: igemm_tp_kernel, materialised dy, replica-aware chunking) equal to fp32
    rounding."""
    from dynaboa_amd import assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    S = 2

    def mk(r):
        o = DB.frame_only_options(inner_step=1)
        o.deferred_metrics = 1
        return DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True), device="cpu")
    frames = [[assets.make_frame(100 * r, 1, seed=22)] for r in range(S)]
    singles = []
    for r in range(S):
        ad = mk(r)
        ad.excute(frames[r], nframes=1)
        st = ad.optimizer.state[ad.model.module.theta]
        singles.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()))
    emu_lib.dyb_set_option(b"rep_split", throughput)
    emu_lib.dyb_set_option(b"tp_min", 2 if throughput else 8)
    emu_lib.dyb_set_option(b"tp_gn_wgs", 16 if throughput else 1024)        # 8 GroupNorm chunks per image instead of up to 128
    try:
        ads = [mk(r) for r in range(S)]
        grp = NS.ReplicaGroup(ads, 1)
        grp.step([frames[r][0] for r in range(S)], 0)
        grp.flush_metrics()
    finally:
        emu_lib.dyb_set_option(b"rep_split", 0)
        emu_lib.dyb_set_option(b"tp_min", 8)
        emu_lib.dyb_set_option(b"tp_gn_wgs", 1024)
    for r in range(S):
        a = ads[r]
        st = a.optimizer.state[a.model.module.theta]
        if not throughput:
            assert torch.equal(a.model.module.theta.detach(), singles[r][0]), r
            assert torch.equal(st["exp_avg"], singles[r][1]) and torch.equal(st["exp_avg_sq"], singles[r][2]), r
        else:
            L = a.model.module._layout1
            g1, g0 = L.unpack(st["exp_avg"]), L.unpack(singles[r][1])
            worst = max(float((g1[k].double() - g0[k].double()).norm() / g0[k].double().norm().clamp_min(1e-30)) for k in g0)
            assert worst < 5e-3, worst
    assert not torch.equal(ads[0].model.module.theta.detach(), ads[1].model.module.theta.detach())


@pytest.mark.skipif(not os.environ.get("DYB_EMU_FULL"), reason="~15 min under the emulator; set DYB_EMU_FULL=1")
def test_full_term_replica_group_on_emulator(emu_lib):
    """The reference's default term set for S = 2 sequence replicas (teacher forward, exemplar pass, per-replica dynamic-BOA gate:
    dyb_stepper_adapt_frames_full) on the emulator against the same sequences adapted one at a time: identical weights, Adam
    state, teacher and step counts (frame 0: no history pair yet)."""
    from dynaboa_amd import assets, benchmark as DB, native_step as NS
    from dynaboa_amd.base_adaptor import synthetic_bundle
    S = 2

    def mk(r):
        o = DB.parser.parse_args([])
        o.inner_step, o.interval, o.optim_steps, o.cos_sim_threshold, o.deferred_metrics = 1, 2, 1, -1.0, 1
        return DB.Adaptor(o, synthetic_bundle(seed=22 + r, identity_pose=False, randomize_norm=True), device="cpu")
    frames = [[assets.make_frame(100 * r, 1, seed=22)] for r in range(S)]
    singles = []
    for r in range(S):
        ad = mk(r)
        ad.excute(frames[r], nframes=1)
        st = ad.optimizer.state[ad.model.module.theta]
        singles.append((ad.model.module.theta.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(st["step"]),
                        ad.teacher.theta.detach().clone(), list(ad.optim_step_record)))
    ads = [mk(r) for r in range(S)]
    grp = NS.ReplicaGroup(ads, 1)
    assert grp.stepper.full
    grp.step([frames[r][0] for r in range(S)], 0)
    grp.flush_metrics()
    for r in range(S):
        a = ads[r]
        st = a.optimizer.state[a.model.module.theta]
        assert int(st["step"]) == singles[r][3] == 2 and list(a.optim_step_record) == singles[r][5]
        assert torch.equal(a.model.module.theta.detach(), singles[r][0]), r
        assert torch.equal(st["exp_avg"], singles[r][1]) and torch.equal(st["exp_avg_sq"], singles[r][2]), r
        assert torch.equal(a.teacher.theta.detach(), singles[r][4]), r
