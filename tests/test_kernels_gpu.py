"""Parity of every HIP kernel at the REAL sizes of the path (all 23 ResNet-50 conv shapes of
SURVEY 8a row 2, the SMPL tables, the 27 M-float arena) through the C ABI on cuda:0, against the
oracle / torch-CPU fp32 ops and the golden vectors."""
import numpy as np
import pytest

import kernel_cases as K
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from backends import GpuBackend
    return GpuBackend()


# (H, W, Cin, Cout, k, stride, pad) - the 23 unique shapes, batch 1
RESNET_SHAPES = [
    (224, 224, 4, 64, 7, 2, 3), (56, 56, 64, 64, 1, 1, 0), (56, 56, 64, 64, 3, 1, 1), (56, 56, 64, 256, 1, 1, 0),
    (56, 56, 256, 64, 1, 1, 0), (56, 56, 256, 128, 1, 1, 0), (56, 56, 128, 128, 3, 2, 1), (28, 28, 128, 512, 1, 1, 0),
    (56, 56, 256, 512, 1, 2, 0), (28, 28, 512, 128, 1, 1, 0), (28, 28, 128, 128, 3, 1, 1), (28, 28, 512, 256, 1, 1, 0),
    (28, 28, 256, 256, 3, 2, 1), (14, 14, 256, 1024, 1, 1, 0), (28, 28, 512, 1024, 1, 2, 0), (14, 14, 1024, 256, 1, 1, 0),
    (14, 14, 256, 256, 3, 1, 1), (14, 14, 1024, 512, 1, 1, 0), (14, 14, 512, 512, 3, 2, 1), (7, 7, 512, 2048, 1, 1, 0),
    (14, 14, 1024, 2048, 1, 2, 0), (7, 7, 2048, 512, 1, 1, 0), (7, 7, 512, 512, 3, 1, 1)]


@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_all_resnet_shapes(be, shape):
    H, W, C, Kc, R, st, pad = shape
    K.case_conv(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc, c_real=3 if C == 4 else None)


@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_operand_pair_all_resnet_shapes(be, shape):
    """The tangent passes' operand pairs - op(a1, b1) + op(a2, b2) in one launch - on every ResNet-50 conv shape."""
    H, W, C, Kc, R, st, pad = shape
    K.case_conv_pair(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc + 1)


@pytest.mark.parametrize("shape", [(56, 56, 64, 64, 3, 1, 1), (28, 28, 512, 1024, 1, 2, 0), (7, 7, 512, 512, 3, 1, 1)])
def test_conv_batch8(be, shape):
    H, W, C, Kc, R, st, pad = shape
    K.case_conv(be, 8, H, W, C, Kc, R, st, pad, seed=5)


@pytest.fixture(params=[2, 1, 3], ids=["pipelined", "phased", "pipelined2"])
def throughput_mode(be, request):
    """Throughput schedule forced on for plain calls (normally: launches covering >= 8 sequence replicas), once with each loop
    form of igemm_tp_kernel (tp_kernel 2 = software-pipelined, the default; 1 = round 2's phase-separated loop; 3 = pipelined with two K-steps of loads in flight)."""
    be.lib.dyb_set_option(b"rep_split", 1)
    be.lib.dyb_set_option(b"tp_min", 1)
    be.lib.dyb_set_option(b"tp_kernel", request.param)
    yield
    be.lib.dyb_set_option(b"rep_split", 0)
    be.lib.dyb_set_option(b"tp_min", 8)
    be.lib.dyb_set_option(b"tp_kernel", 2)


@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_throughput_kernel_all_resnet_shapes(be, throughput_mode, shape):
    """igemm_tp_kernel (128x128 / 64x256 / 256x64 tiles, XCD-contiguous workgroup order) on every ResNet-50 conv shape:
    forward, data gradient (+ addend), weight gradient against torch's convolution."""
    H, W, C, Kc, R, st, pad = shape
    K.case_conv(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc, c_real=3 if C == 4 else None)


@pytest.mark.parametrize("shape", [(56, 56, 64, 64, 3, 1, 1), (28, 28, 512, 1024, 1, 2, 0), (7, 7, 512, 512, 3, 1, 1)])
def test_conv_throughput_kernel_batch8(be, throughput_mode, shape):
    H, W, C, Kc, R, st, pad = shape
    K.case_conv(be, 8, H, W, C, Kc, R, st, pad, seed=5)


@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_throughput_kernel_bf16_all_resnet_shapes(be, throughput_mode, shape):
    """bf16 form of the pipelined throughput kernel on every ResNet-50 conv shape (the stem stays on the latency-form kernel)."""
    H, W, C, Kc, R, st, pad = shape
    K.case_conv(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc, c_real=3 if C == 4 else None, bf16=True)


@pytest.mark.parametrize("tp_grid", [16, 4096])
@pytest.mark.parametrize("shape", [(56, 56, 128, 128, 3, 2, 1), (14, 14, 1024, 2048, 1, 2, 0), (14, 14, 512, 512, 3, 2, 1),
                                   (14, 14, 256, 256, 3, 1, 1)])
def test_conv_throughput_kernel_split_policies(be, throughput_mode, shape, tp_grid):
    """The throughput kernel without a split (tp_grid 16) and with the deepest one the policy allows (4096), incl. the
    stride-2 data gradients by phase class (classes of 4 / 2 / 2 / 1 taps share one split depth)."""
    H, W, C, Kc, R, st, pad = shape
    be.lib.dyb_set_option(b"tp_grid", tp_grid)
    try:
        K.case_conv(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc + tp_grid)
    finally:
        be.lib.dyb_set_option(b"tp_grid", 512)


@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_inkernel_fold_latency_form_all_resnet_shapes(be, shape):
    """Latency form (64x64 tiles) with a counter region in scope on every ResNet-50 conv shape at one image: split launches fold
    in-kernel - last workgroup to arrive on a tile, slabs added in split order - instead of leaving slabs to a fold launch."""
    H, W, C, Kc, R, st, pad = shape
    folds = K.case_conv_inkernel_fold(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc)
    assert folds >= 1, shape                         # every shape splits at least one of its three modes at one image


@pytest.mark.parametrize("tp_grid", [512, 4096])
@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_inkernel_fold_throughput_kernel_all_resnet_shapes(be, throughput_mode, shape, tp_grid):
    """Throughput kernel with a counter region in scope (pipelined forms fold in-kernel; the phased loop and the compact stride-2
    data gradient keep their fold launches) at the policy's split depth and at the deepest one."""
    H, W, C, Kc, R, st, pad = shape
    be.lib.dyb_set_option(b"tp_grid", tp_grid)
    be.lib.dyb_set_option(b"tp_fold", 7)                 # (off by default: measured slower at 16 - 32 sequences, r05 s3)
    try:
        K.case_conv_inkernel_fold(be, 1, H, W, C, Kc, R, st, pad, seed=H + C + Kc + tp_grid)
    finally:
        be.lib.dyb_set_option(b"tp_grid", 512)
        be.lib.dyb_set_option(b"tp_fold", 0)


def test_conv_inkernel_fold_replicas_bit_identical_to_one_sequence(be):
    """Eight replicas in one launch, in-kernel fold on each replica's own counters (region inside the replica arena is not needed for
    plain pointers: the debug entry registers the workspace slices): every replica equals the same convolution run alone, bit for bit,
    whatever the arrival order was (three repetitions)."""
    import ctypes
    import torch
    S, H, C, Kc, R = 8, 14, 256, 256, 3
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(S, 1, H, H, C, generator=g).to(dev)
    w = (torch.randn(S, R, R, C, Kc, generator=g) * 0.05).to(dev)
    dy = torch.randn(S, 1, H, H, Kc, generator=g).to(dev)
    wsb = 64 << 20
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L = be.lib
    L.dyb_set_option(b"rep_split", 1)
    L.dyb_set_option(b"tp_fold", 7)
    try:
        ctr = torch.zeros(1024, dtype=torch.int32, device=dev)
        for mode, ref_shape in ((0, dy), (1, x), (2, w)):
            outs = []
            for rep in range(3):
                out = torch.empty_like(ref_shape)
                assert L.dyb_debug_set_conv_sync(ctr.data_ptr(), 1024) == 0
                try:
                    assert L.dyb_debug_conv_replicas(mode, x.data_ptr(), w.data_ptr(), dy.data_ptr(), out.data_ptr(), S, 1, H, H, C, Kc, R, R, 1, 1,
                                                     ws.data_ptr(), wsb, None) == 0
                finally:
                    L.dyb_debug_set_conv_sync(None, 0)
                torch.cuda.synchronize()
                outs.append(out.clone())
            assert not bool(ctr.any())
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), mode
            # against the fold launch (tree order differs: rounding only)
            ref = torch.empty_like(ref_shape)
            assert L.dyb_debug_conv_replicas(mode, x.data_ptr(), w.data_ptr(), dy.data_ptr(), ref.data_ptr(), S, 1, H, H, C, Kc, R, R, 1, 1,
                                             ws.data_ptr(), wsb, None) == 0
            torch.cuda.synchronize()
            assert float((outs[0] - ref).abs().max() / ref.abs().max()) < 1e-5, mode
    finally:
        L.dyb_set_option(b"rep_split", 0)
        L.dyb_set_option(b"tp_fold", 0)


@pytest.mark.parametrize("cfg", [(1, 14, 14, 256, 256, 3, 1, 512, 3, 1), (1, 56, 56, 64, 64, 3, 1, 256, 3, 1), (1, 28, 28, 128, 128, 3, 2, 512, 3, 1)])
def test_layer_gnstats_from_the_latency_kernels_epilogue(be, cfg):
    ctr = be.zeros((1024,), dtype=np.uint32)
    N, H, W, C, Ka, Ra, sa, Kb, Rb, sb = cfg
    Ha = (H + 2 * (Ra // 2) - Ra) // sa + 1
    assert be.lib.dyb_debug_set_conv_sync(be.ptr(ctr), 1024) == 0
    try:
        r = K.case_layer_gnstats(be, *cfg, seed=sum(cfg))
    finally:
        be.lib.dyb_debug_set_conv_sync(None, 0)
    assert r["nA"] == -(-Ha * Ha // 64) * (Ka // 64), r
    assert not np.asarray(be.host(ctr)).any()


@pytest.mark.parametrize("cfg", [(1, 12544, 64, 1, False, 1), (1, 3136, 256, 1, True, 1), (1, 784, 512, 1, True, 2),
                                 (1, 196, 1024, 0, False, 5), (1, 49, 2048, 1, True, 16), (8, 196, 256, 1, False, 1),
                                 (2, 49, 512, 1, False, 9)])
def test_groupnorm(be, cfg):
    K.case_groupnorm(be, *cfg)


# every GroupNorm of the backbone at one image: (HW, C, relu, mask_from_y, nslabs, with_addend, cap) - block outputs mask from
# the saved activation, layers inside a bottleneck from y, the shortcut branch has no ReLU; slabs / addend as the backward
# chain hands them over; cap 0 = 8192 float4 per workgroup (stem / 56x56x256: 7 chunks, 28x28x512: 4, 56x56x128: 4, ...)
GN_ONEPASS = [(12544, 64, 1, False, 1, False, 0), (3136, 64, 1, True, 4, False, 0), (3136, 256, 1, False, 2, True, 0),
              (3136, 256, 0, False, 1, False, 0), (3136, 128, 1, True, 1, False, 0), (784, 128, 1, True, 2, False, 0),
              (784, 512, 1, False, 1, True, 0), (784, 256, 1, True, 1, False, 0), (196, 256, 1, True, 4, False, 0),
              (196, 1024, 1, False, 4, True, 0), (196, 512, 1, True, 1, False, 0), (49, 512, 1, True, 8, False, 0),
              (49, 2048, 1, False, 8, True, 0), (49, 2048, 0, False, 1, False, 0), (196, 1024, 1, False, 1, False, 2048)]


@pytest.mark.parametrize("cfg", GN_ONEPASS)
def test_groupnorm_onepass(be, cfg):
    print(K.case_groupnorm_onepass(be, *cfg, seed=sum(cfg[:2])))


@pytest.mark.parametrize("shape", RESNET_SHAPES)
def test_conv_gn_bwd_fused_all_resnet_shapes(be, shape):
    H, W, C, Kc, R, st, pad = shape
    K.case_conv_gn_bwd_fused(be, 1, H, W, C, Kc, R, st, pad, relu=0 if (R == 1 and st == 2) else 1, seed=H + C + Kc + 1)


@pytest.mark.parametrize("shape", [(56, 56, 64, 64, 3, 1, 1), (14, 14, 1024, 2048, 1, 2, 0), (7, 7, 512, 512, 3, 1, 1)])
def test_conv_gn_bwd_fused_batch5(be, shape):
    H, W, C, Kc, R, st, pad = shape
    K.case_conv_gn_bwd_fused(be, 5, H, W, C, Kc, R, st, pad, seed=9)


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, planes, stride, downsample   (the distinct bottleneck shapes of ResNet-50 at 224x224)
    (1, 56, 56, 64, 64, 1, True), (1, 56, 56, 256, 64, 1, False), (1, 56, 56, 256, 128, 2, True),
    (1, 28, 28, 512, 128, 1, False), (1, 28, 28, 512, 256, 2, True), (1, 14, 14, 1024, 256, 1, False),
    (1, 14, 14, 1024, 512, 2, True), (1, 7, 7, 2048, 512, 1, False), (4, 14, 14, 1024, 256, 1, False),
    (3, 28, 28, 512, 256, 2, True)])
def test_bottleneck_fused(be, cfg):
    K.case_bottleneck_fused(be, *cfg, seed=sum(int(v) for v in cfg))


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, Ka, Ra, sa, Kb, Rb, sb   (layer pairs of ResNet-50 at 224x224; K4 = the single-launch 1x1 kernel)
    (1, 28, 28, 512, 128, 1, 1, 128, 3, 1),      # layer2 conv1 (K4) -> conv2 (tiled 3x3)
    (1, 28, 28, 128, 512, 1, 1, 128, 1, 1),      # layer2 conv3 (K4) -> next conv1 (K4 with loader GroupNorm)
    (1, 14, 14, 1024, 256, 1, 1, 256, 3, 1),     # layer3 conv1 (K4, 8 K-steps) -> conv2
    (1, 14, 14, 256, 1024, 1, 1, 256, 1, 1),     # layer3 conv3 -> conv1
    (1, 7, 7, 512, 2048, 1, 1, 512, 1, 1),       # layer4 conv3 (K4) -> conv1 (Cin 2048: tiled)
    (1, 56, 56, 256, 512, 1, 2, 128, 1, 1),      # stride-2 downsample (K4, M = 784)
    (1, 56, 56, 64, 256, 1, 1, 64, 1, 1),        # layer1: M = 3136 -> tiled path
    (4, 14, 14, 256, 1024, 1, 1, 256, 1, 1),     # batch 4 -> tiled path
])
def test_layer_gnstats(be, cfg):
    K.case_layer_gnstats(be, *cfg, seed=sum(cfg))


@pytest.mark.parametrize("cfg", [
    # H, W, C (producer channels), K (conv output channels), mask_from_y, with_addend
    (28, 28, 128, 512, True, False),       # layer2 conv3 data gradient + bn2 reduce
    (28, 28, 512, 128, False, True),       # layer2 conv1 data gradient + previous block's bn3 reduce (+ residual edge)
    (14, 14, 256, 1024, True, False),      # layer3 conv3 (8 K-steps)
    (14, 14, 1024, 256, False, True),      # layer3 conv1
    (7, 7, 2048, 512, False, True),        # layer4 conv1
])
def test_dgrad_gn_reduce(be, cfg):
    r = K.case_dgrad_gn_reduce(be, *cfg, seed=sum(int(v) for v in cfg))
    assert r["layouts"][0] != r["layouts"][1]


def test_groupnorm_fold(be):
    K.case_groupnorm_fold(be, 1, 784, 512, 4, True)
    K.case_groupnorm_fold(be, 1, 49, 2048, 36, False)
    K.case_groupnorm_fold(be, 2, 3136, 64, 2, True)


def test_pools(be):
    K.case_pools(be, 2, 112, 112, 64)
    K.case_avgpool(be, 3, 49, 2048)


def test_linear(be):
    K.case_linear(be, 1, 2205, 1024)
    K.case_linear(be, 8, 1024, 1024)
    K.case_linear(be, 3, 1024, 160)


def test_rotations(be):
    K.case_rot6d(be, golden)
    K.case_rotmat_to_aa(be, golden)
    K.case_rodrigues(be)
    K.case_projection(be, B=8)


def test_pa_mpjpe_vs_reference_procrustes(be):
    """dyb_pa_mpjpe (Jacobi-SVD Procrustes on the device) against golden g6 = the reference's reconstruction_error
    (utils/pose_utils.py) incl. the reflection sample, and against numpy on random poses."""
    K.case_pa_mpjpe(be, golden)


def test_perspective_projection_and_gmm_prior(be):
    """The module-level forms the reference's import surface names (utils/geometry.py perspective_projection,
    MaxMixturePrior.forward) on their own kernels."""
    from dynaboa_amd import assets
    K.case_perspective_projection(be)
    K.case_gmm_prior(be, assets.load_gmm_prior())


def test_lbs(be, smpl_tabs):
    K.case_lbs(be, smpl_tabs, B=1, with_dverts=False)
    K.case_lbs(be, smpl_tabs, B=8, with_dverts=True)


def test_lbs_known_answers(be, smpl_tabs):
    """First-principles pins for the third-party (smplx) semantics: identity pose and a rigid root
    rotation."""
    import torch
    from oracle import ref_cpu as O
    from dynaboa_amd._abi import check
    B = 2
    fb, ib, (_a, pf), (_b, pi) = K.smpl_device_tables(be, smpl_tabs)
    betas = (np.random.default_rng(0).standard_normal((B, 10)) * 0.5).astype(np.float32)
    rot = np.tile(np.eye(3, dtype=np.float32), (B, 24, 1, 1))
    R0 = O.smplx_rodrigues(torch.tensor([[0.4, -0.7, 0.3]])).numpy()[0]
    rot[1, 0] = R0
    V, J = be.empty((B, 6890, 3)), be.empty((B, 49, 3))
    saved = be.empty((be.lib.dyb_lbs_saved_floats(B),))
    check(be.lib.dyb_lbs_fwd(pf, pi, be.ptr(be.dev(betas)), 10, be.ptr(be.dev(rot)), be.ptr(V), be.ptr(J), be.ptr(saved), B,
                             be.stream), "lbs")
    v = be.host(V)
    v_shaped = smpl_tabs["v_template"][None] + np.einsum("bl,vcl->bvc", betas, smpl_tabs["shapedirs"])
    assert np.abs(v[0] - v_shaped[0]).max() < 2e-6
    J0 = (smpl_tabs["J_regressor"] @ v_shaped[1])[0]
    assert np.abs(v[1] - ((v_shaped[1] - J0) @ R0.T + J0)).max() < 5e-6


def test_frame_losses(be):
    from dynaboa_amd import assets
    K.case_frame_losses(be, golden, assets.load_gmm_prior())


def test_optim_full_arena(be):
    K.case_optim(be, n=26_977_504)


def test_conv_linearity_full_size(be):
    """Size-independent property at the largest layer: conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    from dynaboa_amd._abi import check
    rng = np.random.default_rng(3)
    N, H, W, C, Kc = 8, 56, 56, 64, 256
    w = be.dev((rng.standard_normal((1, 1, C, Kc)) / 8).astype(np.float32))
    x1, x2 = rng.standard_normal((N, H, W, C)).astype(np.float32), rng.standard_normal((N, H, W, C)).astype(np.float32)
    wsb = be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, Kc, 1, 1, 1, 0)
    ws = be.empty((max(wsb, 16) // 4,))
    outs = []
    for x in (x1, x2, 2.5 * x1 + x2):
        y = be.empty((N, H, W, Kc))
        check(be.lib.dyb_conv2d_nhwc_fwd(be.ptr(be.dev(x)), be.ptr(w), be.ptr(y), N, H, W, C, Kc, 1, 1, 1, 0, be.ptr(ws), wsb,
                                         be.stream), "conv")
        outs.append(be.host(y))
    assert np.abs(outs[2] - (2.5 * outs[0] + outs[1])).max() < 1e-4 * np.abs(outs[2]).max()


@pytest.mark.parametrize("onepass", [2, 1, 0])
def test_hmr_engine_one_image_throughput_schedule_vs_latency_schedule(be, ckpt_rand, onepass):
    """One image per launch replica (the benchmarked configuration): the engine's gradients under the throughput schedule - one-pass
    GroupNorm backward for every layer (2), for one-workgroup slabs only (1), two-launch reduce + apply (0) - against its own
    latency schedule (pinned to the reference module by test_hmr_engine_vs_reference_module)."""
    print(K.case_hmr_engine_schedules(be, ckpt_rand, {"rep_split": 1, "tp_min": 1, "tp_gn_onepass": onepass}))


@pytest.mark.parametrize("k4_batch", [1, 0])
def test_hmr_engine_vs_reference_module(be, ckpt_rand, k4_batch):
    """Whole engine at batch 2 against the reference module's golden: with the single-launch 1x1 kernels at batch > 1
    (k4_batch=1, the default) and with the tiled kernels only."""
    be.lib.dyb_set_option(b"k4_batch", k4_batch)
    try:
        e = K.case_hmr_engine(be, golden, ckpt_rand)
    finally:
        be.lib.dyb_set_option(b"k4_batch", 1)
    print(e)


def test_hmr_engine_throughput_schedule_vs_reference_module(be, ckpt_rand):
    """The engine's throughput schedule (what launches covering >= 8 sequence replicas use: dy materialised once per layer by
    the GroupNorm-backward apply kernel, plain data- / weight-gradient convolutions, no single-launch 1x1 kernels), forced on
    for a plain batch-2 call, against the reference module's golden g3."""
    be.lib.dyb_set_option(b"rep_split", 1)
    be.lib.dyb_set_option(b"tp_min", 1)
    try:
        e = K.case_hmr_engine(be, golden, ckpt_rand)
    finally:
        be.lib.dyb_set_option(b"rep_split", 0)
        be.lib.dyb_set_option(b"tp_min", 8)
    print(e)


@pytest.mark.parametrize("k4_batch", [1, 0])
def test_batched_pairs_both_dispatches(be, k4_batch):
    """The batch > 1 layer pairs / data-gradient + reduce pairs / bottlenecks through both dispatches (single-launch 1x1
    kernels with per-image tiles, or tiled conv + statistics)."""
    be.lib.dyb_set_option(b"k4_batch", k4_batch)
    try:
        r = K.case_layer_gnstats(be, 4, 14, 14, 256, 1024, 1, 1, 256, 1, 1, seed=11)
        assert (r["nA"] == 7 * 32) == bool(k4_batch)             # 196 pixels -> 7 row tiles per image x 32 column tiles
        K.case_layer_gnstats(be, 8, 28, 28, 512, 128, 1, 1, 128, 3, 1, seed=12)
        K.case_layer_gnstats(be, 3, 7, 7, 512, 2048, 1, 1, 512, 1, 1, seed=13)
        K.case_dgrad_gn_reduce(be, 14, 14, 1024, 256, False, True, seed=14, N=8)
        K.case_dgrad_gn_reduce(be, 28, 28, 128, 512, True, False, seed=15, N=3)
        K.case_bottleneck_fused(be, 8, 14, 14, 1024, 256, 1, False, seed=16)
        K.case_bottleneck_fused(be, 2, 28, 28, 512, 256, 2, True, seed=17)
    finally:
        be.lib.dyb_set_option(b"k4_batch", 1)


def test_aux_loss_terms(be):
    K.case_aux_terms(be, B=1, seed=5)
    K.case_aux_terms(be, B=8)
    K.case_aux_terms(be, B=16, seed=3)


@pytest.mark.parametrize("cfg", [(1, 3136, 64, 1, False), (1, 784, 512, 1, True, True), (2, 196, 1024, 0, False), (1, 49, 2048, 1, True),
                                 (1, 12544, 64, 1, False, True), (16, 196, 1024, 1, True)])
def test_groupnorm_tangent_kernels(be, cfg):
    """Forward tangent of GroupNorm(+ReLU)(+residual) and the tangent of its backward (exact Hessian-vector product building
    blocks) against torch's forward-over-reverse in float64, at ResNet-50 layer sizes."""
    K.case_gn_jvp(be, *cfg)


@pytest.mark.parametrize("cfg", [(1, 3136, 64, 1, False), (1, 784, 512, 1, True, True), (2, 196, 1024, 0, False), (1, 49, 2048, 1, True),
                                 (1, 12544, 64, 1, False, True), (16, 196, 1024, 1, True)])
def test_groupnorm_tangent_kernels_one_launch(be, cfg):
    """The same with sums and apply as ONE launch each (a slab's row chunks meet on an arrival counter inside the launch)."""
    K.case_gn_jvp(be, *cfg, onepass=True)


@pytest.mark.parametrize("side", [True, False])
def test_hmr_exact_hessian_vector_product(be, ckpt_rand, side):
    """dyb_hmr_jvp_forward / _backward: tangent of the regressor state and every tensor of H v against torch differentiating
    the oracle twice (CPU); with the side stream (the pairs' off-chain halves beside the chain) and with everything in line."""
    print(K.case_hmr_hvp(be, ckpt_rand, side=side))


@pytest.mark.parametrize("cfg", [
    (1, 7, 7, 128, 256, 3, 1, 1),      # 49-pixel reduction (ragged last K-step), 128x128 tiles
    (1, 12, 12, 128, 128, 1, 1, 0),    # 1x1
    (2, 10, 10, 64, 64, 3, 2, 1),      # Cout = 64: the 256x64 form, stride 2, batch 2
    (1, 14, 14, 64, 128, 1, 2, 0),     # 1x1 stride 2 (downsample)
])
def test_conv_weight_gradient_writes_fast_weights(be, cfg):
    """"fuse_fast": the throughput-form weight gradient with a weight-update scope in force (kernel_cases.case_conv_wgrad_update) -
    unsplit (tp_grid 1): p_next = p_cur - lr * g from the epilogue, the gradient buffer untouched; split (tp_grid 4096 where the shape
    allows a split): the scope is ignored and the plain gradient arrives."""
    N, H, W, C, Kc, R, st, pad = cfg
    be.lib.dyb_set_option(b"rep_split", 1)
    be.lib.dyb_set_option(b"tp_min", 1)
    try:
        be.lib.dyb_set_option(b"tp_grid", 1)
        assert K.case_conv_wgrad_update(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg)) == 1
        assert K.case_conv_wgrad_adam(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg) + 2) == 1       # "fuse_adam": Adam from the accumulators
        be.lib.dyb_set_option(b"tp_grid", 4096)
        K.case_conv_wgrad_update(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg) + 1)
        K.case_conv_wgrad_adam(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg) + 3)
    finally:
        be.lib.dyb_set_option(b"tp_grid", 512)
        be.lib.dyb_set_option(b"rep_split", 0)
        be.lib.dyb_set_option(b"tp_min", 8)


@pytest.mark.parametrize("sync", [0, 1], ids=["fold_launch_or_unsplit", "in_kernel_fold"])
@pytest.mark.parametrize("cfg", [
    (1, 7, 7, 128, 256, 3, 1, 1),      # deep reduction over taps: split by the latency policy
    (1, 14, 14, 64, 64, 1, 1, 0),      # small 1x1
    (1, 20, 20, 4, 64, 7, 2, 3),       # stem
    (2, 10, 10, 64, 64, 3, 2, 1),      # stride 2, batch 2
])
def test_conv_weight_gradient_writes_fast_weights_latency_form(be, cfg, sync):
    """"fuse_fast" for ONE sequence (latency form, igemm_mfma_kernel): the finished weight-gradient tile - unsplit, folded by the fold launch
    (splitk_reduce_kernel: addend + scale * sum of slabs) or, with a counter region in scope, folded in-kernel by the last workgroup to
    arrive - leaves p_next = p_cur - lr * g; the gradient buffer stays untouched."""
    import numpy as np
    N, H, W, C, Kc, R, st, pad = cfg
    ctr = be.zeros((4096,), dtype=np.uint32)
    if sync:
        assert be.lib.dyb_debug_set_conv_sync(be.ptr(ctr), 4096) == 0
    try:
        assert K.case_conv_wgrad_update(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg) + sync) == 1
    finally:
        be.lib.dyb_debug_set_conv_sync(None, 0)
    assert not np.asarray(be.host(ctr)).any()
