"""Kernel logic on CPU: the product .hip sources compiled for the host (tests/emu) and run
work-item by work-item, compared with the oracle.  Small shapes; the real sizes are covered by
tests/test_kernels_gpu.py on the MI355X."""
import numpy as np
import pytest

import kernel_cases as K
from conftest import golden


@pytest.fixture(scope="module")
def be():
    from backends import EmuBackend
    return EmuBackend()


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, K, R, stride, pad
    (1, 8, 8, 64, 64, 1, 1, 0),        # 1x1, exact tiles
    (1, 7, 7, 64, 128, 3, 1, 1),       # 3x3, M=49 ragged, split-K engaged
    (2, 10, 10, 64, 64, 3, 2, 1),      # 3x3 stride 2, batch 2
    (1, 8, 8, 128, 64, 1, 2, 0),       # 1x1 stride 2 (downsample)
    (1, 20, 20, 4, 64, 7, 2, 3),       # stem: 7x7 s2, Cin padded 3->4, K=196 ragged
])
def test_conv(be, cfg):
    N, H, W, C, Kc, R, st, pad = cfg
    K.case_conv(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg), c_real=3 if C == 4 else None)


@pytest.mark.parametrize("cfg", [
    (1, 8, 8, 64, 64, 1, 1, 0),        # 1x1, exact tiles
    (1, 7, 7, 64, 128, 3, 1, 1),       # 3x3, ragged rows, split-K over the doubled K loop
    (2, 10, 10, 64, 64, 3, 2, 1),      # stride 2, batch 2
    (1, 20, 20, 4, 64, 7, 2, 3),       # 7x7 s2 with a ragged K tile per pair (196 = 6 x 32 + 4)
])
def test_conv_operand_pair(be, cfg):
    """One launch for op(a1, b1) + op(a2, b2) (the tangent passes' pairs): forward, data gradient + addend, weight gradient."""
    print(K.case_conv_pair(be, *cfg, seed=sum(cfg)))


def test_conv_operand_pair_declines_what_it_does_not_cover(be):
    """-1 for a bad mode or a missing second pair, -3 where the throughput schedule is in force or the option is off - never a silent
    single-pair result."""
    N, H, W, C, Kc, R, st, pad = 1, 8, 8, 64, 64, 1, 1, 0
    x, w, y = be.dev(np.zeros((N, H, W, C), np.float32)), be.dev(np.zeros((R, R, C, Kc), np.float32)), be.empty((N, H, W, Kc))
    wsb = be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, Kc, R, R, st, pad)
    ws = be.empty((max(wsb, 16) // 4,))
    g = (N, H, W, C, Kc, R, R, st, pad, be.ptr(ws), wsb, be.stream)
    call = lambda mode, a2, b2: be.lib.dyb_debug_conv_pair(mode, be.ptr(x), be.ptr(w), a2, b2, be.ptr(y), None, *g)
    assert call(0, be.ptr(x), be.ptr(w)) == 0
    assert call(3, be.ptr(x), be.ptr(w)) == -1
    assert call(0, None, be.ptr(w)) == -1
    be.lib.dyb_set_option(b"conv_pair", 0)
    try:
        assert call(0, be.ptr(x), be.ptr(w)) == -3
    finally:
        be.lib.dyb_set_option(b"conv_pair", 1)
    be.lib.dyb_set_option(b"rep_split", 1)
    be.lib.dyb_set_option(b"tp_min", 1)             # throughput schedule for plain calls: the pair has no form there
    try:
        assert call(0, be.ptr(x), be.ptr(w)) == -3
    finally:
        be.lib.dyb_set_option(b"rep_split", 0)
        be.lib.dyb_set_option(b"tp_min", 8)
    assert be.lib.dyb_gn_jvp_sync_words(0) == 0 and be.lib.dyb_gn_jvp_sync_words(2) == 2 * 4 + 1


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


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, K, R, stride, pad          tile forms of igemm_tp_kernel reached (fwd / dgrad / wgrad)
    (1, 12, 12, 128, 128, 1, 1, 0),    # 128x128 everywhere, ragged M = 144
    (1, 7, 7, 128, 256, 3, 1, 1),      # M = 49: 64x256 (fwd, dgrad); wgrad 128x128 with a 49-pixel reduction (ragged K-step)
    (2, 10, 10, 64, 64, 3, 2, 1),      # Cout = 64: 256x64 forms, stride 2, batch 2, split-K
    (1, 8, 8, 128, 64, 1, 2, 0),       # 1x1 stride 2 (downsample)
    (1, 20, 20, 4, 64, 7, 2, 3),       # stem: forward on the C4 form of the pipelined kernel (per-thread taps; 64x64 kernel for "phased"), weight gradient 256x64
    (2, 21, 19, 4, 64, 7, 2, 3),       # stem, odd H != W, batch 2: 2 x 11 x 10 = 220 rows (ragged 256-row tile), K = 196 = 12 steps + 4 of 16
    (1, 16, 16, 4, 32, 5, 1, 2),       # Cin = 4 with a 5x5 stride-1 filter (S = 5: a wrap every K-step), Cout 32: K = 100
    (1, 9, 9, 4, 64, 4, 1, 1),         # S = 4 = taps per K-step: (r, s) -> (r + 1, s) every step
    (1, 4, 4, 64, 128, 3, 1, 1),       # 4x4 map: a K-step of the weight gradient spans whole images (pipelined form falls back to the pixel-walk loop)
    (2, 5, 6, 32, 128, 3, 1, 1),       # 5x6 map, batch 2: one wrap per K-step of the branch-free pixel walk, twice
])
def test_conv_throughput_kernel(be, throughput_mode, cfg):
    """igemm_tp_kernel (the 128x128-class tiles the throughput schedule runs) through the plain entry points: forward,
    data gradient (+ residual addend) and weight gradient against torch's convolution."""
    N, H, W, C, Kc, R, st, pad = cfg
    K.case_conv(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg), c_real=3 if C == 4 else None)


@pytest.mark.parametrize("tp_grid", [512, 8])
@pytest.mark.parametrize("cfg", [
    (1, 14, 14, 64, 128, 1, 2, 0),     # 1x1 stride 2: one phase class has the tap, three write zeros (+ addend); class rows 49 -> 64x256 form
    (1, 14, 14, 64, 128, 3, 2, 1),     # 3x3 stride 2: classes with 4 / 2 / 2 / 1 taps
    (3, 14, 14, 32, 64, 3, 2, 1),      # batch 3, Cin 32 (256x64 form for the data gradient)
    (1, 9, 9, 16, 32, 3, 2, 1),        # odd size: classes of unequal row counts
    (1, 16, 12, 16, 16, 3, 2, 1),      # H != W
])
def test_conv_throughput_data_gradient_by_phase_class(be, throughput_mode, cfg, tp_grid):
    """Stride-2 data gradient of igemm_tp_kernel: rows enumerated by ((h+pad)&1, (w+pad)&1) class, each class's K loop over its
    own taps; with a deep split (tp_grid 512 on these small shapes) and without one (8)."""
    be.lib.dyb_set_option(b"tp_grid", tp_grid)
    try:
        K.case_conv(be, *cfg, seed=7)
    finally:
        be.lib.dyb_set_option(b"tp_grid", 512)


@pytest.mark.parametrize("tp_grid", [512, 64])
@pytest.mark.parametrize("cfg", [
    (1, 12, 12, 128, 128, 3, 1, 1),    # 128x128 tiles, ragged M, K loop of 72 steps split deep
    (1, 7, 7, 128, 256, 3, 1, 1),      # 64x256 form
    (2, 10, 10, 64, 64, 3, 2, 1),      # 256x64 forms, stride-2 data gradient by phase class
    (1, 14, 14, 64, 128, 3, 2, 1),     # stride 2, classes with 4 / 2 / 2 / 1 taps (some splits of a class are empty)
    (1, 14, 14, 64, 128, 1, 2, 0),     # 1x1 stride 2: the compact form keeps its scatter fold
])
def test_conv_throughput_kernel_inkernel_fold(be, throughput_mode, cfg, tp_grid):
    """Split-K launches of igemm_tp_kernel with a counter region in scope: the last workgroup to arrive on a tile folds the tile's
    slabs in split order, adds the addend and writes the result (pipelined loop forms; the phased loop keeps the fold launch)."""
    be.lib.dyb_set_option(b"tp_grid", tp_grid)
    be.lib.dyb_set_option(b"tp_fold", 7)                 # (off by default: measured slower at 16 - 32 sequences, r05 s3)
    try:
        folds = K.case_conv_inkernel_fold(be, *cfg, seed=sum(cfg))
    finally:
        be.lib.dyb_set_option(b"tp_grid", 512)
        be.lib.dyb_set_option(b"tp_fold", 0)
    import ctypes
    v = ctypes.c_int(0)
    be.lib.dyb_get_option(b"tp_kernel", ctypes.byref(v))
    if v.value >= 2 and tp_grid == 512 and cfg[5] == 3:
        assert folds >= 1, "no launch took the in-kernel fold"
    if v.value < 2:
        assert folds == 0


@pytest.mark.parametrize("cfg", [(1, 12, 12, 128, 128, 1, 1, 0), (1, 7, 7, 128, 256, 3, 1, 1), (2, 10, 10, 64, 64, 3, 2, 1)])
def test_conv_throughput_kernel_bf16(be, throughput_mode, cfg):
    """bf16 form of the pipelined throughput kernel (operands rounded to bf16 in registers, v_mfma_f32_32x32x16_bf16, fp32
    accumulate): equal to an fp32 convolution of the bf16-rounded operands up to summation order."""
    K.case_conv(be, *cfg, seed=sum(cfg), bf16=True)


def test_conv_timing_table_and_probe(be, throughput_mode):
    """Measurement aids: the per-shape table of a timing scope names the throughput kernel's launches; the phase probe writes
    one record per wave of the matching launch (clocks are 0 on the emulator; the K-step count is real)."""
    import ctypes
    assert be.lib.dyb_conv_timing_begin(64) == 0
    probe = be.dev(np.zeros(64 * 4 * 8 * 2, np.float32))      # 64 workgroups x 4 waves x 8 64-bit words
    be.lib.dyb_conv_probe_set(be.ptr(probe), 64, 0, 12, 128, 128, 1)
    try:
        K.case_conv(be, 1, 12, 12, 128, 128, 1, 1, 0, seed=3)
    finally:
        be.lib.dyb_conv_probe_set(None, 0, 0, 0, 0, 0, 0)
        ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
        assert be.lib.dyb_conv_timing_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)) == 0
    assert n.value == 3 and fl.value > 0
    need = be.lib.dyb_conv_timing_table(None, 0)
    buf = ctypes.create_string_buffer(int(need))
    be.lib.dyb_conv_timing_table(buf, need)
    kinds = sorted(line.split(",")[0] for line in buf.value.decode().strip().splitlines()[1:])
    assert kinds == ["t", "u", "v"], kinds
    rec = be.host(probe).view(np.uint64).reshape(64, 4, 8)
    used = rec[rec[:, 0, 6] != 0]
    assert len(used) >= 1 and (used[..., 6] == used[0, 0, 6]).all()


@pytest.mark.parametrize("cfg", [(1, 49, 64, 1, True, 1), (2, 30, 128, 1, False, 3), (1, 12, 2048, 0, False, 1),
                                 (1, 300, 64, 1, True, 2)])
def test_groupnorm(be, cfg):
    K.case_groupnorm(be, *cfg)


@pytest.mark.parametrize("cfg", [
    # HW, C, relu, mask_from_y, nslabs, with_addend, cap
    (49, 64, 1, False, 1, False, 0),        # 7x7, one chunk, mask from the saved activation
    (49, 2048, 1, True, 1, False, 0),       # widest layer (128 float4 columns per group: two waves carry the columns), mask from y
    (36, 128, 0, False, 1, False, 0),       # no ReLU (the shortcut branch's GroupNorm)
    (50, 64, 1, True, 3, True, 0),          # split-K slabs + residual-edge addend folded on the way in
    (60, 64, 1, False, 2, True, 64),        # 4 row chunks per slab meet on the counter (cap 64 float4 per workgroup)
    (45, 256, 1, True, 1, False, 128),      # 6 chunks, ragged last chunk
])
def test_groupnorm_onepass(be, cfg):
    K.case_groupnorm_onepass(be, *cfg, seed=sum(cfg[:2]))


def test_groupnorm_fold(be):
    K.case_groupnorm_fold(be, 1, 49, 64, 5, True)
    K.case_groupnorm_fold(be, 2, 20, 512, 2, False)
    K.case_groupnorm_fold(be, 1, 30, 128, 1, True)


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, K, R, stride, pad, relu
    (1, 7, 7, 64, 64, 3, 1, 1, 1),       # 3x3, ragged M, ReLU mask
    (3, 6, 6, 64, 128, 1, 1, 0, 1),      # batch 3: pixel tiles straddle images (per-image coefficients)
    (2, 8, 8, 64, 64, 3, 2, 1, 1),       # stride 2
    (1, 8, 8, 128, 256, 1, 2, 0, 0),     # downsample flavour: no ReLU, dm aliases dout
])
def test_conv_gn_bwd_fused(be, cfg):
    K.case_conv_gn_bwd_fused(be, *cfg, seed=sum(cfg))


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, planes, stride, downsample
    (1, 6, 6, 64, 16, 1, False),      # identity shortcut
    (2, 8, 8, 64, 32, 2, True),       # stride-2 block with downsample, batch 2
    (3, 5, 5, 32, 16, 1, True),       # ragged tiles straddling images
])
def test_bottleneck_fused(be, cfg):
    K.case_bottleneck_fused(be, *cfg, seed=sum(int(v) for v in cfg))


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, Ka, Ra, sa, Kb, Rb, sb
    (1, 6, 6, 64, 128, 1, 1, 128, 1, 1),       # K4 -> K4 (the second with the producer's GroupNorm in its loader)
    (1, 9, 9, 128, 128, 1, 2, 256, 1, 1),      # K4 with stride 2 (downsample flavour), ragged 25-pixel map
    (1, 6, 6, 64, 128, 1, 1, 64, 3, 1),        # K4 -> tiled 3x3 (partials of the tile kind consumed by the tiled loader)
    (1, 6, 6, 64, 64, 3, 1, 128, 1, 1),        # tiled 3x3 -> K4
    (2, 6, 6, 64, 128, 1, 1, 128, 1, 1),       # batch 2: tiled path for the same shapes
])
def test_layer_gnstats(be, cfg):
    r = K.case_layer_gnstats(be, *cfg, seed=sum(cfg))
    assert r["nA"] > 0 and r["nB"] > 0


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, Ka, Ra, sa, Kb, Rb, sb       (one image: the statistics of both layers' outputs leave with the conv tiles)
    (1, 12, 12, 64, 64, 3, 1, 128, 1, 1),      # groups of 16 columns: a wave tile holds all four; then groups of 32
    (1, 9, 9, 64, 256, 1, 1, 512, 1, 1),       # groups of 64 / 128 columns (one group per wave tile, several tiles per group), ragged 81 rows
    (1, 16, 16, 128, 128, 3, 2, 64, 3, 1),     # stride 2, then the 256x64 tile form
    (1, 7, 7, 256, 512, 1, 1, 256, 3, 1),      # 49 rows: the 64x256 form
])
def test_layer_gnstats_from_the_throughput_kernels_epilogue(be, cfg):
    """One image, no K split (tp_grid 1): igemm_tp_kernel's forward epilogue leaves one (sum, sum of squares) record per wave tile and
    no statistics launch follows; the next layer's loader and the apply kernel fold those records (test body as test_layer_gnstats)."""
    be.lib.dyb_set_option(b"rep_split", 1)
    be.lib.dyb_set_option(b"tp_min", 1)
    be.lib.dyb_set_option(b"tp_grid", 1)
    try:
        r = K.case_layer_gnstats(be, *cfg, seed=sum(cfg))
        N, H, W, C, Ka, Ra, sa, Kb, Rb, sb = cfg
        Ha = (H + 2 * (Ra // 2) - Ra) // sa + 1
        tm = 256 if Ka <= 64 else (64 if Ha * Ha <= 64 else 128)
        tn = 64 if Ka <= 64 else (256 if Ha * Ha <= 64 else 128)
        assert r["nA"] == -(-Ha * Ha // tm) * (tm // 64) * -(-Ka // tn) * (tn // 64), r     # the records are the wave tiles, not row chunks
        be.lib.dyb_set_option(b"tp_gn_fuse_stats", 0)
        r0 = K.case_layer_gnstats(be, *cfg, seed=sum(cfg))
        assert r0["nA"] != r["nA"] or Ha * Ha <= 64
    finally:
        be.lib.dyb_set_option(b"rep_split", 0)
        be.lib.dyb_set_option(b"tp_min", 8)
        be.lib.dyb_set_option(b"tp_grid", 512)
        be.lib.dyb_set_option(b"tp_gn_fuse_stats", 1)


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, K, R, stride, pad     latency form (64x64 tiles): the epilogue goes through LDS; deep split-K at these sizes
    (1, 14, 14, 64, 64, 3, 1, 1),      # 196 rows: ragged last tile
    (2, 7, 7, 128, 64, 3, 1, 1),       # batch 2 (rows span images)
    (1, 12, 12, 128, 256, 1, 1, 0),    # 1x1, four column tiles
    (1, 10, 10, 64, 128, 3, 2, 1),     # stride 2
    (1, 20, 20, 4, 64, 7, 2, 3),       # stem shape (Cin padded to 4)
])
def test_conv_inkernel_fold_latency_form(be, cfg):
    """igemm_mfma_kernel with a counter region in scope: split launches fold in-kernel (the last workgroup to arrive on a tile adds
    the slabs in split order), the unsplit ones store their tile through the same LDS-transposed epilogue."""
    folds = K.case_conv_inkernel_fold(be, *cfg, seed=sum(cfg))
    assert folds >= 1, "no launch of this case split its K loop"
    be.lib.dyb_set_option(b"lat_fold", 0)
    try:
        assert K.case_conv_inkernel_fold(be, *cfg, seed=sum(cfg)) == 0
    finally:
        be.lib.dyb_set_option(b"lat_fold", 1)


@pytest.mark.parametrize("cfg", [
    # N, H, W, C, Ka, Ra, sa, Kb, Rb, sb       (one image, tiled path for both layers)
    (1, 12, 12, 64, 64, 3, 1, 128, 3, 1),      # groups of 16 columns (four per tile), then 32 (two per tile); 144 rows: ragged
    (1, 9, 9, 64, 256, 3, 1, 512, 3, 1),       # groups of 64 (one per tile) and 128 columns (two tiles per group)
    (1, 16, 16, 128, 128, 3, 2, 64, 3, 1),     # stride 2
])
def test_layer_gnstats_from_the_latency_kernels_epilogue(be, cfg):
    """One image on the latency schedule with a counter region in scope: the conv launch - split or not - leaves one statistics record
    per workgroup tile (from the folded tile where K was split) and no statistics launch follows."""
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
    r0 = K.case_layer_gnstats(be, *cfg, seed=sum(cfg))          # no region: unsplit launches still carry their statistics
    assert r0["nA"] > 0


def _random_conv_cfgs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        R = int(rng.choice([1, 3]))
        st = int(rng.choice([1, 2]))
        pad = 1 if R == 3 else 0
        H, W = int(rng.integers(3, 12)), int(rng.integers(3, 12))
        if (H + 2 * pad - R) // st + 1 < 1 or (W + 2 * pad - R) // st + 1 < 1:
            continue
        out.append((int(rng.integers(1, 4)), H, W, int(rng.choice([16, 32, 64, 128])), int(rng.choice([16, 32, 64, 128])), R, st, pad))
    return out


@pytest.mark.parametrize("cfg", _random_conv_cfgs(8, 123))
def test_conv_random_shapes(be, cfg):
    """Seeded random geometry (non-square maps, batch 1-3, ragged tiles in every dimension, K tails): the three plain
    gather flavours and the GroupNorm-fused gradients against torch autograd."""
    N, H, W, C, Kc, R, st, pad = cfg
    K.case_conv(be, N, H, W, C, Kc, R, st, pad, seed=sum(cfg))
    K.case_conv_gn_bwd_fused(be, N, H, W, C, Kc, R, st, pad, relu=1, seed=sum(cfg) + 1)


def test_conv_is_linear_in_both_operands(be):
    """Size-independent property: conv(a*x1 + x2, w) == a*conv(x1, w) + conv(x2, w), same in w (split-K included)."""
    rng = np.random.default_rng(9)
    N, H, W, C, Kc, R, st, pad = 1, 9, 7, 64, 64, 3, 1, 1
    x1, x2 = (rng.standard_normal((N, H, W, C)).astype(np.float32) for _ in range(2))
    w1, w2 = ((rng.standard_normal((R, R, C, Kc)) / 24).astype(np.float32) for _ in range(2))
    wsb = max(be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, Kc, R, R, st, pad), 16)
    ws = be.empty((wsb // 4,))

    def conv(x, w):
        y = be.empty((N, H, W, Kc))
        K.check(be.lib.dyb_conv2d_nhwc_fwd(be.ptr(be.dev(x)), be.ptr(be.dev(w)), be.ptr(y), N, H, W, C, Kc, R, R, st, pad,
                                           be.ptr(ws), wsb, be.stream), "conv")
        return be.host(y).copy()
    a = 1.75
    assert K.rel_err(conv(a * x1 + x2, w1), a * conv(x1, w1) + conv(x2, w1)) < 2e-6
    assert K.rel_err(conv(x1, a * w1 + w2), a * conv(x1, w1) + conv(x1, w2)) < 2e-6


@pytest.mark.parametrize("cfg", [
    # H, W, C (producer / conv input channels), K (conv output channels), mask_from_y, with_addend
    (6, 6, 128, 128, True, False),      # conv3-flavour: producer bn2, mask recomputed from y, no residual
    (5, 7, 256, 128, False, True),      # conv1-flavour: producer bn3 (stored activation + residual), residual-edge addend, ragged 35-pixel map
    (4, 4, 128, 256, True, True),
])
def test_dgrad_gn_reduce(be, cfg):
    r = K.case_dgrad_gn_reduce(be, *cfg, seed=sum(int(v) for v in cfg))
    assert r["layouts"][0] != r["layouts"][1]            # the one-launch path really ran (tile-count layout)


def test_k4_batched_paths(be):
    """k4_batch=1 (default): the single-launch 1x1 kernels at batch > 1 - tiles never straddle images, per-image
    partial records / coefficients - forward pair and backward pair against torch."""
    be.lib.dyb_set_option(b"k4_batch", 1)
    r = K.case_layer_gnstats(be, 3, 5, 5, 128, 128, 1, 1, 256, 1, 1, seed=77)            # ragged 25-pixel maps, 3 images
    assert r["nA"] == 1 * 4 and r["nB"] == 1 * 8                                          # tiles per image x column tiles
    r = K.case_dgrad_gn_reduce(be, 5, 7, 256, 128, False, True, seed=78, N=2)
    assert r["layouts"][0] == (2, 8)                                                      # 35 pixels -> 2 row tiles per image
    K.case_dgrad_gn_reduce(be, 4, 4, 128, 256, True, False, seed=79, N=3)


def test_pools(be):
    K.case_pools(be, 2, 12, 12, 64)
    K.case_avgpool(be, 2, 49, 128)


def test_linear(be):
    K.case_linear(be, 1, 157 + 99, 40)      # ragged in_features -> padded stride
    K.case_linear(be, 5, 512, 160)          # batch tile > 4


def test_rotations(be):
    K.case_rot6d(be, golden)
    K.case_rotmat_to_aa(be, golden)
    K.case_rodrigues(be)
    K.case_projection(be)


def test_perspective_projection_and_gmm_prior(be):
    """The module-level forms the reference's import surface names (utils/geometry.py perspective_projection,
    MaxMixturePrior.forward) on their own kernels."""
    from dynaboa_amd import assets
    K.case_perspective_projection(be)
    K.case_gmm_prior(be, assets.load_gmm_prior())


def test_lbs(be, smpl_tabs):
    K.case_lbs(be, smpl_tabs, B=2, with_dverts=True)
    K.case_lbs(be, smpl_tabs, B=1, with_dverts=False)


def test_frame_losses(be):
    from dynaboa_amd import assets
    K.case_frame_losses(be, golden, assets.load_gmm_prior())


def test_pa_mpjpe(be):
    K.case_pa_mpjpe(be, golden)


def test_optim(be):
    K.case_optim(be, n=4096)


@pytest.mark.slow
@pytest.mark.parametrize("onepass", [2, 0], ids=["gn_onepass", "gn_two_launch"])
def test_hmr_engine_throughput_schedule_vs_reference_module(be, ckpt_rand, onepass):
    """The throughput schedule (materialised dy, plain gradient convolutions; used by launches covering >= 8 sequence
    replicas) forced on for a plain call: the whole engine against the reference module's golden g3 - with the one-pass
    GroupNorm backward (default; the stem's and layer1's slabs are 7 row chunks meeting on a counter) and with the
    two-launch reduce + apply."""
    be.lib.dyb_set_option(b"rep_split", 1)
    be.lib.dyb_set_option(b"tp_min", 1)
    be.lib.dyb_set_option(b"tp_gn_onepass", onepass)
    try:
        K.case_hmr_engine(be, golden, ckpt_rand)
    finally:
        be.lib.dyb_set_option(b"rep_split", 0)
        be.lib.dyb_set_option(b"tp_min", 8)
        be.lib.dyb_set_option(b"tp_gn_onepass", 2)


@pytest.mark.slow
def test_hmr_engine_one_image_throughput_schedule_with_onepass_groupnorm_backward(be, ckpt_rand):
    """One image per launch replica is where the one-pass GroupNorm backward runs (the benchmarked configuration): the engine's
    gradients under the throughput schedule against its own latency schedule; the stem and layer1 slabs are cut into 7 row
    chunks whose workgroups meet on a counter."""
    print(K.case_hmr_engine_schedules(be, ckpt_rand, {"rep_split": 1, "tp_min": 1, "tp_gn_onepass": 2}))


@pytest.mark.slow
def test_hmr_engine_throughput_schedule_by_batch(be, ckpt_rand):
    """The throughput schedule selected by the batch size of a single-sequence launch (switch tp_batch_min, 8 by default):
    the whole engine at batch 2 against the reference module's golden g3."""
    be.lib.dyb_set_option(b"tp_batch_min", 2)
    try:
        K.case_hmr_engine(be, golden, ckpt_rand)
    finally:
        be.lib.dyb_set_option(b"tp_batch_min", 8)


@pytest.mark.skipif(__import__("os").environ.get("DYB_EMU_FULL") != "1", reason="opt-in (DYB_EMU_FULL=1): minutes on the emulator")
@pytest.mark.parametrize("k4_batch", [0, 1])
def test_hmr_engine_batch2_vs_reference_module(be, ckpt_rand, k4_batch):
    """The whole engine (forward + backward, batch 2) against the reference module's golden g3 on the emulator; with
    k4_batch=1 the single-launch 1x1 kernels run with per-image tiles / partial records (experimental path)."""
    be.lib.dyb_set_option(b"k4_batch", k4_batch)
    try:
        K.case_hmr_engine(be, golden, ckpt_rand)
    finally:
        be.lib.dyb_set_option(b"k4_batch", 1)


@pytest.mark.parametrize("cfg", [(1, 9, 9, 32, 64, 3, 1, 1), (2, 8, 8, 64, 32, 1, 1, 0), (1, 12, 12, 16, 64, 3, 2, 1), (1, 6, 6, 128, 128, 1, 2, 0)])
def test_conv_bf16_variant(be, cfg):
    """bf16 matrix-core variant of the tiled conv (forward / data gradient / weight gradient): equal to an fp32 convolution
    of the bf16-rounded operands up to summation order."""
    K.case_conv(be, *cfg, seed=sum(cfg), bf16=True)


def test_aux_loss_terms(be):
    K.case_aux_terms(be, B=3)
    K.case_aux_terms(be, B=1, seed=5)


@pytest.mark.parametrize("cfg", [(1, 49, 64, 1, True), (2, 30, 128, 1, False), (1, 12, 2048, 0, False), (3, 20, 256, 0, True),
                                 (1, 3136, 64, 1, True, True), (2, 200, 512, 1, False, True)])      # 13 / 7 row chunks per slab, ragged tails
def test_groupnorm_tangent_kernels(be, cfg):
    """Forward tangent of GroupNorm(+ReLU)(+residual) and the tangent of its backward (the building blocks of the exact
    Hessian-vector product) against torch's forward-over-reverse in float64."""
    K.case_gn_jvp(be, *cfg)


@pytest.mark.parametrize("cfg", [(1, 49, 64, 1, True), (3, 20, 256, 0, True), (1, 3136, 64, 1, True, True), (2, 200, 512, 1, False, True)])
def test_groupnorm_tangent_kernels_one_launch(be, cfg):
    """The same with sums and apply as ONE launch each (the chunks of a slab meet on an arrival counter; the group's channel sums wait
    for every image's chunks)."""
    K.case_gn_jvp(be, *cfg, onepass=True)


def test_side_stream_schedules_hold_under_adversarial_stream_order(be, ckpt_rand):
    """The first-order backward (weight gradients on the side stream) and the exact Hessian-vector product's two tangent passes (the
    off-chain halves of every pair on the side stream) in the emulator's lazy stream mode, drained chain-first and side-stream-first:
    bit-identical to the in-line run, i.e. every cross-stream dependency is covered by an event wait."""
    print(K.case_stream_order(be, ckpt_rand))


@pytest.mark.slow
@pytest.mark.parametrize("B", [1, pytest.param(2, marks=pytest.mark.skipif(__import__("os").environ.get("DYB_EMU_FULL") != "1",
                                                                            reason="opt-in (DYB_EMU_FULL=1): +75 s"))])
def test_hmr_exact_hessian_vector_product(be, ckpt_rand, B):
    """The tangent passes through the whole network (exact H v, forward-over-reverse) against torch differentiating the oracle
    twice: tangent of the regressor state and every tensor of H v; batch 1 and 2."""
    print(K.case_hmr_hvp(be, ckpt_rand, B=B))


@pytest.mark.skipif(__import__("os").environ.get("DYB_EMU_FULL") != "1", reason="opt-in (DYB_EMU_FULL=1): +40 s; the option paths, both off by default")
def test_hmr_exact_hessian_vector_product_one_launch_groupnorm_tangents(be, ckpt_rand, monkeypatch):
    """The same with DYB_HVP_GN_ONEPASS=1 (the engine's per-layer arrival counters, zeroed per pass) and with the operand pairs off
    (two launches per tangent pair, second halves collected in hv2)."""
    monkeypatch.setenv("DYB_HVP_GN_ONEPASS", "1")
    be.lib.dyb_set_option(b"conv_pair", 0)
    try:
        print(K.case_hmr_hvp(be, ckpt_rand, B=1))
    finally:
        be.lib.dyb_set_option(b"conv_pair", 1)


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
