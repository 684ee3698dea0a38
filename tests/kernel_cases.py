"""Per-kernel parity cases: the C ABI (through a backend from tests/backends.py) against the CPU
oracle / plain torch-CPU fp32 ops on the same seeded inputs.  Shapes are parameters so the same
bodies serve the emulator (tiny) and the GPU box (the real ResNet-50 / SMPL sizes)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from conftest import rel_err
from dynaboa_amd._abi import check
from oracle import ref_cpu as O

TOL = 2e-4        # fp32 accumulation-order differences; north_star tolerance is 1e-3 relative


def _rng(seed):
    return np.random.default_rng(seed)


# ---------------------------------------------------------------------------------------- conv
def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (nearest even) -> fp32: what the bf16 matrix-core variant does to an operand tile as it is staged."""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def case_conv(be, N, H, W, C, K, R, stride, pad, seed=0, c_real=None, bf16=False):
    """bf16=True: the entry points run with the option "bf16" on (operands rounded to bf16 when staged, fp32 accumulate);
    the reference then convolves the ROUNDED operands in fp32, so only the summation order differs."""
    rng = _rng(seed)
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    if c_real is not None:
        x[..., c_real:] = 0
    w = (rng.standard_normal((R, R, C, K)) / np.sqrt(R * R * C)).astype(np.float32)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = rng.standard_normal((N, Ho, Wo, K)).astype(np.float32)
    add = rng.standard_normal((N, H, W, C)).astype(np.float32)
    if bf16:
        be.lib.dyb_set_option(b"bf16", 1)
        try:
            xr, wr, dyr = bf16_round(x), bf16_round(w), bf16_round(dy)
            e = _conv_check(be, N, H, W, C, K, R, stride, pad, x, w, dy, add, xr, wr, dyr)
        finally:
            be.lib.dyb_set_option(b"bf16", 0)
        return e
    return _conv_check(be, N, H, W, C, K, R, stride, pad, x, w, dy, add, x, w, dy)


def _conv_check(be, N, H, W, C, K, R, stride, pad, x, w, dy, add, xr, wr, dyr):
    xt = torch.from_numpy(xr).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wt = torch.from_numpy(wr).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    dy_ref_in = dyr
    yt = F.conv2d(xt, wt, stride=stride, padding=pad)
    gx, gw = torch.autograd.grad(yt, [xt, wt], torch.from_numpy(dy_ref_in).permute(0, 3, 1, 2))
    y_ref = yt.detach().permute(0, 2, 3, 1).numpy()
    dx_ref = gx.permute(0, 2, 3, 1).numpy() + add
    dw_ref = gw.permute(2, 3, 1, 0).numpy()

    wsb = be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad)
    ws = be.empty((max(wsb, 16) // 4,))
    dx_, dw_, y_ = be.empty(x.shape), be.empty(w.shape), be.empty(dy.shape)
    X, Wt, DY, ADD = be.dev(x), be.dev(w), be.dev(dy), be.dev(add)
    check(be.lib.dyb_conv2d_nhwc_fwd(be.ptr(X), be.ptr(Wt), be.ptr(y_), N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb,
                                     be.stream), "conv fwd")
    check(be.lib.dyb_conv2d_nhwc_dgrad(be.ptr(DY), be.ptr(Wt), be.ptr(dx_), be.ptr(ADD), N, H, W, C, K, R, R, stride, pad,
                                       be.ptr(ws), wsb, be.stream), "conv dgrad")
    check(be.lib.dyb_conv2d_nhwc_wgrad(be.ptr(X), be.ptr(DY), be.ptr(dw_), N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb,
                                       be.stream), "conv wgrad")
    e = dict(fwd=rel_err(be.host(y_), y_ref), dgrad=rel_err(be.host(dx_), dx_ref), wgrad=rel_err(be.host(dw_), dw_ref))
    assert max(e.values()) < TOL, e
    return e


def case_conv_wgrad_update(be, N, H, W, C, K, R, stride, pad, lr=0.37, seed=0):
    """"fuse_fast" (round 6): with a weight-update scope set, an UNSPLIT throughput-form weight gradient leaves p_cur - lr * g in p_next
    instead of g in dw (igemm_tp.inc epilogue: addend + out_scale * acc).  Against torch's gradient; dw must stay untouched.  The caller
    has put the library on the throughput schedule with tp_grid small enough for nsplit == 1.  -> launches that took the fused form."""
    rng = _rng(seed)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    dy = rng.standard_normal((N, Ho, Wo, K)).astype(np.float32)
    p_cur = rng.standard_normal((R, R, C, K)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    wt = torch.zeros(K, C, R, R, requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv2d(xt, wt, stride=stride, padding=pad), [wt], torch.from_numpy(dy).permute(0, 3, 1, 2))
    dw_ref = gw.permute(2, 3, 1, 0).numpy()
    wsb = be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad)
    ws = be.empty((max(wsb, 16) // 4,))
    marker = np.full(p_cur.shape, 7.25, np.float32)
    X, DY, P0 = be.dev(x), be.dev(dy), be.dev(p_cur)
    dw_, p1_ = be.dev(marker), be.dev(np.zeros_like(p_cur))
    check(be.lib.dyb_debug_set_wgrad_update(be.ptr(dw_), p_cur.size * 4, be.ptr(P0), be.ptr(p1_), lr), "set_wgrad_update")
    try:
        check(be.lib.dyb_conv2d_nhwc_wgrad(be.ptr(X), be.ptr(DY), be.ptr(dw_), N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb, be.stream),
              "conv wgrad (update scope)")
        be.sync()
        fused = int(be.lib.dyb_debug_wgrad_update_spans())
    finally:
        be.lib.dyb_debug_set_wgrad_update(None, 0, None, None, 0.0)
    if fused:
        assert np.array_equal(be.host(dw_), marker), "the gradient buffer was written although the update was fused"
        e = rel_err(be.host(p1_), p_cur - np.float32(lr) * dw_ref)
    else:
        e = rel_err(be.host(dw_), dw_ref)
    assert e < TOL, e
    return fused


def case_conv_wgrad_adam(be, N, H, W, C, K, R, stride, pad, seed=0, t_step=3):
    """"fuse_adam" (round 6): with an Adam scope set, an UNSPLIT throughput-form weight gradient applies torch.optim.Adam's step to theta /
    exp_avg / exp_avg_sq in place from its accumulators (igemm_tp.inc).  Against torch.optim.Adam itself on torch's gradient (step
    t_step: bias corrections as the stepper computes them); the gradient buffer must stay untouched.  -> launches that took the form."""
    rng = _rng(seed)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    dy = (rng.standard_normal((N, Ho, Wo, K)) * 0.1).astype(np.float32)
    theta = rng.standard_normal((R, R, C, K)).astype(np.float32)
    m0 = (rng.standard_normal(theta.shape) * 0.05).astype(np.float32)
    v0 = (rng.random(theta.shape) * 0.01).astype(np.float32)
    lr, b1, b2, eps = 3e-3, 0.5, 0.9, 1e-8
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
    wt = torch.zeros(K, C, R, R, requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv2d(xt, wt, stride=stride, padding=pad), [wt], torch.from_numpy(dy).permute(0, 3, 1, 2))
    g = gw.permute(2, 3, 1, 0).contiguous()
    # torch.optim.Adam at step t_step with these moments (single-tensor formula)
    p = torch.nn.Parameter(torch.from_numpy(theta.copy()))
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, foreach=False)
    opt.state[p] = dict(step=torch.tensor(float(t_step - 1)), exp_avg=torch.from_numpy(m0.copy()), exp_avg_sq=torch.from_numpy(v0.copy()))
    p.grad = g.clone()
    opt.step()
    sc = np.array([lr / (1.0 - b1 ** t_step), np.sqrt(1.0 - b2 ** t_step)], np.float32)
    wsb = be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad)
    ws = be.empty((max(wsb, 16) // 4,))
    marker = np.full(theta.shape, 7.25, np.float32)
    X, DY, TH, M, V, SC = be.dev(x), be.dev(dy), be.dev(theta), be.dev(m0), be.dev(v0), be.dev(sc)
    dw_ = be.dev(marker)
    check(be.lib.dyb_debug_set_wgrad_adam(be.ptr(dw_), theta.size * 4, be.ptr(TH), be.ptr(M), be.ptr(V), be.ptr(SC), b1, b2, eps), "set_wgrad_adam")
    try:
        check(be.lib.dyb_conv2d_nhwc_wgrad(be.ptr(X), be.ptr(DY), be.ptr(dw_), N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb, be.stream),
              "conv wgrad (Adam scope)")
        be.sync()
        fused = int(be.lib.dyb_debug_wgrad_update_spans())
    finally:
        be.lib.dyb_debug_set_wgrad_update(None, 0, None, None, 0.0)
    if fused:
        assert np.array_equal(be.host(dw_), marker), "the gradient buffer was written although Adam was fused"
        st = opt.state[p]
        e = dict(theta=rel_err(be.host(TH) - theta, p.detach().numpy() - theta), m=rel_err(be.host(M), st["exp_avg"].numpy()),
                 v=rel_err(be.host(V), st["exp_avg_sq"].numpy()))
        assert max(e.values()) < 5 * TOL, e
    else:
        assert rel_err(be.host(dw_), g.numpy()) < TOL
        assert np.array_equal(be.host(TH), theta)
    return fused


def case_conv_inkernel_fold(be, N, H, W, C, K, R, stride, pad, seed=0):
    """The same three checks with a counter region in scope: a launch that splits K lets the workgroup that arrives last on a tile add
    the tile's slabs itself (igemm_tp.inc / igemm_conv.hip epilogues) instead of leaving them to a fold launch.  The counters must be
    back at zero afterwards; returns how many launches folded in-kernel."""
    import ctypes
    ctr = be.zeros((1024,), dtype=np.uint32)
    before = ctypes.c_int(0)
    be.lib.dyb_get_option(b"stat_folds", ctypes.byref(before))
    check(be.lib.dyb_debug_set_conv_sync(be.ptr(ctr), 1024), "set_conv_sync")
    try:
        case_conv(be, N, H, W, C, K, R, stride, pad, seed=seed, c_real=3 if C == 4 else None)
        be.sync()
    finally:
        be.lib.dyb_debug_set_conv_sync(None, 0)
    after = ctypes.c_int(0)
    be.lib.dyb_get_option(b"stat_folds", ctypes.byref(after))
    assert not np.asarray(be.host(ctr)).any(), "arrival counters not back at zero"
    return after.value - before.value


def case_conv_pair(be, N, H, W, C, K, R, stride, pad, seed=0):
    """dyb_debug_conv_pair (the tangent passes' operand pairs: one launch, one K loop over both pairs) against torch:
    forward conv(x1, w1) + conv(x2, w2); data gradient dgrad(dy1, w1) + dgrad(dy2, w2) + addend; weight gradient wgrad(x1, dy1) +
    wgrad(x2, dy2)."""
    rng = _rng(seed)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    xs = [rng.standard_normal((N, H, W, C)).astype(np.float32) for _ in range(2)]
    ws_ = [(rng.standard_normal((R, R, C, K)) / np.sqrt(R * R * C)).astype(np.float32) for _ in range(2)]
    dys = [rng.standard_normal((N, Ho, Wo, K)).astype(np.float32) for _ in range(2)]
    add = rng.standard_normal((N, H, W, C)).astype(np.float32)
    y_ref, dx_ref, dw_ref = 0, add.copy(), 0
    for x, w, dy in zip(xs, ws_, dys):
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        wt = torch.from_numpy(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
        yt = F.conv2d(xt, wt, stride=stride, padding=pad)
        gx, gw = torch.autograd.grad(yt, [xt, wt], torch.from_numpy(dy).permute(0, 3, 1, 2))
        y_ref = y_ref + yt.detach().permute(0, 2, 3, 1).numpy()
        dx_ref = dx_ref + gx.permute(0, 2, 3, 1).numpy()
        dw_ref = dw_ref + gw.permute(2, 3, 1, 0).numpy()
    wsb = be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad)
    ws = be.empty((max(wsb, 16) // 4,))
    X, Wd, DY, ADD = [be.dev(x) for x in xs], [be.dev(w) for w in ws_], [be.dev(d) for d in dys], be.dev(add)
    y_, dx_, dw_ = be.empty(dys[0].shape), be.empty(xs[0].shape), be.empty(ws_[0].shape)
    g = (N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb, be.stream)
    check(be.lib.dyb_debug_conv_pair(0, be.ptr(X[0]), be.ptr(Wd[0]), be.ptr(X[1]), be.ptr(Wd[1]), be.ptr(y_), None, *g), "pair fwd")
    check(be.lib.dyb_debug_conv_pair(1, be.ptr(DY[0]), be.ptr(Wd[0]), be.ptr(DY[1]), be.ptr(Wd[1]), be.ptr(dx_), be.ptr(ADD), *g), "pair dgrad")
    check(be.lib.dyb_debug_conv_pair(2, be.ptr(X[0]), be.ptr(DY[0]), be.ptr(X[1]), be.ptr(DY[1]), be.ptr(dw_), None, *g), "pair wgrad")
    e = dict(fwd=rel_err(be.host(y_), y_ref), dgrad=rel_err(be.host(dx_), dx_ref), wgrad=rel_err(be.host(dw_), dw_ref))
    assert max(e.values()) < TOL, e
    return e


# ---------------------------------------------------------------------------------------- groupnorm
def case_groupnorm(be, N, HW, C, relu, with_res, nslabs, seed=1):
    rng = _rng(seed)
    y = (rng.standard_normal((N, HW, C)) * 1.5 + 0.3).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(C)).astype(np.float32)
    res = rng.standard_normal((N, HW, C)).astype(np.float32) if with_res else None
    dout = rng.standard_normal((N, HW, C)).astype(np.float32)
    yt = torch.from_numpy(y).permute(0, 2, 1).reshape(N, C, HW, 1).contiguous().requires_grad_(True)
    gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    o = F.group_norm(yt, 4, gt, bt, 1e-5)
    rt = None
    if with_res:
        rt = torch.from_numpy(res).permute(0, 2, 1).reshape(N, C, HW, 1).contiguous().requires_grad_(True)
        o = o + rt
    if relu:
        o = F.relu(o)
    ins = [yt, gt, bt] + ([rt] if with_res else [])
    grads = torch.autograd.grad(o, ins, torch.from_numpy(dout).permute(0, 2, 1).reshape(N, C, HW, 1))
    out_ref = o.detach().reshape(N, C, HW).permute(0, 2, 1).numpy()
    dy_ref = grads[0].reshape(N, C, HW).permute(0, 2, 1).numpy()

    wsb = be.lib.dyb_groupnorm_workspace_bytes(N, HW, C)
    ws = be.empty((wsb // 4,))
    if nslabs > 1:           # split y into random slabs that sum to it
        parts = rng.standard_normal((nslabs - 1, N, HW, C)).astype(np.float32)
        slabs = np.concatenate([parts, (y - parts.sum(0))[None]], 0)
        y = slabs.sum(0).astype(np.float32)      # exactly what the kernel will form (up to order)
        S, Y = be.dev(slabs), be.empty((N, HW, C))
    else:
        S, Y = None, be.dev(y)
    OUT, ST = be.empty((N, HW, C)), be.empty((N, 4, 2))
    check(be.lib.dyb_groupnorm_fwd(be.ptr(S), nslabs, be.ptr(Y), be.ptr(be.dev(gamma)), be.ptr(be.dev(beta)),
                                   be.ptr(be.dev(res)) if with_res else None, be.ptr(OUT), be.ptr(ST), N, HW, C, relu,
                                   be.ptr(ws), wsb, be.stream), "gn fwd")
    e = dict(out=rel_err(be.host(OUT), out_ref))
    if nslabs > 1:
        e["y"] = rel_err(be.host(Y), y)
    DYB, DRES = be.empty((N, HW, C)), (be.empty((N, HW, C)) if with_res else None)
    DG, DB = be.empty((C,)), be.empty((C,))
    check(be.lib.dyb_groupnorm_bwd(be.ptr(be.dev(dout)), be.ptr(OUT), be.ptr(Y), be.ptr(ST), be.ptr(be.dev(gamma)),
                                   be.ptr(DYB), be.ptr(DRES), be.ptr(DG), be.ptr(DB), N, HW, C, relu, be.ptr(ws), wsb,
                                   be.stream), "gn bwd")
    e["dy"] = rel_err(be.host(DYB), dy_ref)
    e["dgamma"] = rel_err(be.host(DG), grads[1].numpy())
    e["dbeta"] = rel_err(be.host(DB), grads[2].numpy())
    if with_res:
        e["dres"] = rel_err(be.host(DRES), grads[3].reshape(N, C, HW).permute(0, 2, 1).numpy())
    assert max(e.values()) < 5e-4, e
    return e


# ---------------------------------------------------------------------------------------- pooling
def case_pools(be, N, H, W, C, seed=2):
    rng = _rng(seed)
    x = np.maximum(rng.standard_normal((N, H, W, C)), 0).astype(np.float32)      # post-ReLU like the model
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yt = F.max_pool2d(xt, 3, 2, 1)
    Ho, Wo = yt.shape[2], yt.shape[3]
    dy = rng.standard_normal((N, Ho, Wo, C)).astype(np.float32)
    (gx,) = torch.autograd.grad(yt, xt, torch.from_numpy(dy).permute(0, 3, 1, 2))
    Y, IDX, DX = be.empty((N, Ho, Wo, C)), be.zeros((N, Ho, Wo, C // 4), np.int32), be.empty(x.shape)
    check(be.lib.dyb_maxpool3x3s2_fwd(be.ptr(be.dev(x)), be.ptr(Y), be.ptr(IDX), N, H, W, C, be.stream), "maxpool fwd")
    check(be.lib.dyb_maxpool3x3s2_bwd(be.ptr(be.dev(dy)), be.ptr(IDX), be.ptr(DX), N, H, W, C, be.stream), "maxpool bwd")
    e = dict(max_fwd=rel_err(be.host(Y), yt.detach().permute(0, 2, 3, 1).numpy()))
    # ties only occur between zeros, whose gradient the preceding ReLU kills: compare where x > 0
    m = x > 0
    e["max_bwd"] = rel_err(be.host(DX)[m], gx.permute(0, 2, 3, 1).numpy()[m])
    # image repack
    img = rng.standard_normal((N, 3, H, W)).astype(np.float32)
    X4 = be.empty((N, H, W, 4))
    check(be.lib.dyb_nchw3_to_nhwc4(be.ptr(be.dev(img)), be.ptr(X4), N, H, W, be.stream), "repack")
    ref4 = np.concatenate([img.transpose(0, 2, 3, 1), np.zeros((N, H, W, 1), np.float32)], -1)
    e["repack"] = float(np.abs(be.host(X4) - ref4).max())
    assert max(e.values()) < 1e-6, e
    return e


def case_avgpool(be, N, HW, C, seed=3):
    rng = _rng(seed)
    x = rng.standard_normal((N, HW, C)).astype(np.float32)
    ld = C + 160
    d0, d1 = be.zeros((N, ld)), be.zeros((N, ld))
    arr, p = be.ptr_array([d0, d1])
    check(be.lib.dyb_avgpool_fwd(be.ptr(be.dev(x)), p, 2, ld, N, HW, C, be.stream), "avgpool fwd")
    e = dict(fwd=rel_err(be.host(d0)[:, :C], x.mean(1)), fwd2=rel_err(be.host(d1)[:, :C], x.mean(1)))
    g = rng.standard_normal((N, ld)).astype(np.float32)
    DX = be.empty((N, HW, C))
    check(be.lib.dyb_avgpool_bwd(be.ptr(be.dev(g)), ld, be.ptr(DX), N, HW, C, be.stream), "avgpool bwd")
    e["bwd"] = rel_err(be.host(DX), np.broadcast_to(g[:, None, :C] / HW, (N, HW, C)))
    assert max(e.values()) < 1e-5, e
    return e


# ---------------------------------------------------------------------------------------- linear
def case_linear(be, B, I_real, O, seed=4):
    rng = _rng(seed)
    I = (I_real + 3) // 4 * 4
    w = np.zeros((O, I), np.float32)
    w[:, :I_real] = rng.standard_normal((O, I_real)) / np.sqrt(I_real)
    bias = rng.standard_normal(O).astype(np.float32)
    x = np.zeros((B, I), np.float32)
    x[:, :I_real] = rng.standard_normal((B, I_real))
    res = rng.standard_normal((B, O)).astype(np.float32)
    Y = be.empty((B, O))
    Wd = be.dev(w)
    check(be.lib.dyb_linear_fwd(be.ptr(be.dev(x)), I, be.ptr(Wd), I, be.ptr(be.dev(bias)), be.ptr(be.dev(res)), O, be.ptr(Y), O,
                                B, I, O, be.stream), "linear fwd")
    e = dict(fwd=rel_err(be.host(Y), x @ w.T + bias + res))
    dy = rng.standard_normal((B, O)).astype(np.float32)
    wsb = be.lib.dyb_linear_bwd_workspace_bytes(B, I, O)
    ws = be.empty((wsb // 4,))
    split = I - 8 if I > 64 else I
    A0 = rng.standard_normal((B, split)).astype(np.float32)
    DA = be.dev(A0)
    addB = rng.standard_normal((B, I - split)).astype(np.float32) if split < I else None
    DB = be.empty((B, I - split)) if split < I else None
    check(be.lib.dyb_linear_bwd_dx(be.ptr(be.dev(dy)), O, be.ptr(Wd), I, B, I, O, be.ptr(DA), split, 1, split, be.ptr(DB),
                                   I - split, be.ptr(be.dev(addB)) if addB is not None else None, I - split, be.ptr(ws), wsb,
                                   be.stream), "linear dx")
    dx_ref = dy @ w
    e["dx_acc"] = rel_err(be.host(DA), A0 + dx_ref[:, :split])
    if split < I:
        e["dx_tail"] = rel_err(be.host(DB), dx_ref[:, split:] + addB)
    # rank-(T*B) outer product
    T = 3
    dys = [rng.standard_normal((B, O)).astype(np.float32) for _ in range(T)]
    xs = [rng.standard_normal((B, I)).astype(np.float32) for _ in range(T)]
    dyb, xb = [be.dev(a) for a in dys], [be.dev(a) for a in xs]
    ka, pa = be.ptr_array(dyb)
    kb, pb = be.ptr_array(xb)
    ld1 = (ctypes.c_int * T)(*([O] * T))
    ld2 = (ctypes.c_int * T)(*([I] * T))
    DW, DBI = be.empty((O, I)), be.empty((O,))
    check(be.lib.dyb_linear_bwd_dw(pa, ctypes.cast(ld1, ctypes.c_void_p), pb, ctypes.cast(ld2, ctypes.c_void_p), T, B, I, O,
                                   be.ptr(DW), I, be.ptr(DBI), be.stream), "linear dw")
    e["dw"] = rel_err(be.host(DW), sum(a.T @ b for a, b in zip(dys, xs)))
    e["db"] = rel_err(be.host(DBI), sum(a.sum(0) for a in dys))
    assert max(e.values()) < TOL, e
    return e


# ---------------------------------------------------------------------------------------- rotations
def case_rot6d(be, golden):
    g = golden("g1_geometry.npz")
    x6 = g["x6"]
    B = x6.shape[0]
    R = be.empty((B * 24, 3, 3))
    X = be.dev(x6)
    check(be.lib.dyb_rot6d_fwd(be.ptr(X), 144, be.ptr(R), B, be.stream), "rot6d fwd")
    e = dict(fwd=rel_err(be.host(R), g["rot6d_R"]))
    DX = be.empty((B, 144))
    check(be.lib.dyb_rot6d_bwd(be.ptr(X), 144, be.ptr(be.dev(g["rot6d_w"])), be.ptr(DX), 144, B, be.stream), "rot6d bwd")
    e["bwd"] = rel_err(be.host(DX), g["rot6d_gx"])
    assert max(e.values()) < 1e-5, e
    return e


def case_rodrigues(be, n=200, seed=12):
    rng = _rng(seed)
    aa = (rng.standard_normal((n, 3)) * 0.8).astype(np.float32)
    aa[0] = 0
    aa[1] = [1e-5, -2e-5, 3e-6]
    R = be.empty((n, 3, 3))
    check(be.lib.dyb_rodrigues_fwd(be.ptr(be.dev(aa)), be.ptr(R), n, be.stream), "rodrigues")
    ref = O.smplx_rodrigues(torch.from_numpy(aa)).numpy()
    e = float(np.abs(be.host(R) - ref).max())
    assert e < 2e-6, e
    return dict(err=e)


def case_rotmat_to_aa(be, golden):
    g = golden("g1_geometry.npz")
    R = g["rodrigues_R"]
    n = R.shape[0]
    AA, DR = be.empty((n, 3)), be.empty((n, 3, 3))
    Rd = be.dev(R)
    check(be.lib.dyb_rotmat_to_aa_fwd(be.ptr(Rd), be.ptr(AA), n, be.stream), "r2aa fwd")
    check(be.lib.dyb_rotmat_to_aa_bwd(be.ptr(Rd), be.ptr(be.dev(g["r2aa_w"])), be.ptr(DR), n, be.stream), "r2aa bwd")
    aa = be.host(AA)
    np.testing.assert_allclose(aa, g["r2aa_out"], rtol=1e-4, atol=2e-6)
    ref = g["r2aa_gR"]
    ok = np.isfinite(ref).all(axis=(1, 2))
    ok[:2] = False                      # theta = 0 and theta ~ 1e-4: singular in the reference's own autograd
    got = be.host(DR)
    # rows near theta ~ pi amplify fp32 noise in both implementations: compare with a per-row scale
    for i in np.nonzero(ok)[0]:
        assert np.abs(got[i] - ref[i]).max() < 2e-3 * max(1.0, np.abs(ref[i]).max()), (i, got[i], ref[i])
    return dict(fwd=float(np.abs(aa - g["r2aa_out"]).max()))


# ---------------------------------------------------------------------------------------- SMPL
def smpl_device_tables(be, tab):
    from dynaboa_amd import constants as C
    Jr = tab["J_regressor"].astype(np.float64)
    f = [tab["v_template"], tab["shapedirs"].reshape(-1, 10), tab["posedirs"], np.ascontiguousarray(tab["lbs_weights"].T),
         (Jr @ tab["v_template"].astype(np.float64)).astype(np.float32),
         np.einsum("jv,vcl->jcl", Jr, tab["shapedirs"].astype(np.float64)).reshape(72, 10).astype(np.float32),
         tab["J_regressor_extra"]]
    i = [tab["parents"].astype(np.int32), np.array(C.VERTEX_JOINT_IDS, np.int32), np.array(C.JOINT_MAP_49, np.int32)]
    fb = [be.dev(a) for a in f]
    ib = [be.dev(a, np.int32) for a in i]
    return fb, ib, be.ptr_array(fb), be.ptr_array(ib)


def case_lbs(be, tab, B, seed=5, with_dverts=True):
    rng = _rng(seed)
    T = O.smpl_tables_to_torch(tab)
    betas = torch.from_numpy((rng.standard_normal((B, 10)) * 0.5).astype(np.float32)).requires_grad_(True)
    rot = O.smplx_rodrigues(torch.from_numpy((rng.standard_normal((B * 24, 3)) * 0.3).astype(np.float32))).view(B, 24, 3, 3)
    rot = rot.detach().clone().requires_grad_(True)
    verts, j49 = O.smpl_forward(T, betas, rot[:, 1:], rot[:, 0:1], pose2rot=False)
    dj = torch.from_numpy(rng.standard_normal((B, 49, 3)).astype(np.float32))
    dv = torch.from_numpy((rng.standard_normal((B, 6890, 3)) * 0.01).astype(np.float32))
    loss = (j49 * dj).sum() + ((verts * dv).sum() if with_dverts else 0.0)
    gb, gr = torch.autograd.grad(loss, [betas, rot])

    fb, ib, (_ka, pf), (_kb, pi) = smpl_device_tables(be, tab)
    V, J49 = be.empty((B, 6890, 3)), be.empty((B, 49, 3))
    saved = be.empty((be.lib.dyb_lbs_saved_floats(B),))
    BE, ROT = be.dev(betas.detach().numpy()), be.dev(rot.detach().numpy())
    check(be.lib.dyb_lbs_fwd(pf, pi, be.ptr(BE), 10, be.ptr(ROT), be.ptr(V), be.ptr(J49), be.ptr(saved), B, be.stream), "lbs fwd")
    e = dict(verts=rel_err(be.host(V), verts.detach().numpy()), joints=rel_err(be.host(J49), j49.detach().numpy()))
    wsb = be.lib.dyb_lbs_bwd_workspace_bytes(B)
    ws = be.empty((wsb // 4,))
    DR, DBt = be.empty((B, 24, 3, 3)), be.empty((B, 10))
    check(be.lib.dyb_lbs_bwd(pf, pi, be.ptr(ROT), be.ptr(saved), be.ptr(be.dev(dj.numpy())),
                             be.ptr(be.dev(dv.numpy())) if with_dverts else None, be.ptr(DR), be.ptr(DBt), 10, B, be.ptr(ws),
                             wsb, be.stream), "lbs bwd")
    e["drot"] = rel_err(be.host(DR), gr.numpy())
    e["dbetas"] = rel_err(be.host(DBt), gb.numpy())
    assert max(e.values()) < TOL, e
    return e


# ---------------------------------------------------------------------------------------- losses
def case_frame_losses(be, golden, gmm):
    """Against golden g4 (the reference's own loss code on the oracle's SMPL joints)."""
    g = golden("g4_losses.npz")
    B = g["shape"].shape[0]
    rot = O.smplx_rodrigues(torch.from_numpy(g["aa"])).view(B, 24, 3, 3).numpy()
    joints = g["s3d"]
    # reference gradients w.r.t. (rot, shape, cam) include the path through SMPL; the kernel returns
    # the partial derivatives w.r.t. its own inputs, so compare against oracle autograd with joints
    # treated as an input
    rt = torch.from_numpy(rot).requires_grad_(True)
    st = torch.from_numpy(g["shape"]).requires_grad_(True)
    ct = torch.from_numpy(g["cam"]).requires_grad_(True)
    jt = torch.from_numpy(joints).requires_grad_(True)
    kp = torch.from_numpy(g["kp"])
    gm = {k: torch.from_numpy(v) for k, v in gmm.items()}
    l2d, lsh, lpo = O.kp2d_loss(O.projection(ct, jt), kp), O.shape_prior(st), O.pose_prior(rt, gm)
    tot = 10.0 * l2d + 2e-6 * lsh + 1e-4 * lpo
    gr, gs, gc, gj = torch.autograd.grad(tot, [rt, st, ct, jt])
    L = be.empty((4,))
    DR, DS, DC, DJ = be.empty((B, 24, 9)), be.empty((B, 10)), be.empty((B, 3)), be.empty((B, 49, 3))
    ws = be.empty((B * 4,))
    logw = np.log(gmm["nll_weights"]).astype(np.float32).reshape(-1)
    check(be.lib.dyb_frame_losses(be.ptr(be.dev(rot)), be.ptr(be.dev(g["shape"])), 10, be.ptr(be.dev(g["cam"])), 3,
                                  be.ptr(be.dev(joints)), be.ptr(be.dev(g["kp"])), be.ptr(be.dev(gmm["means"])),
                                  be.ptr(be.dev(gmm["precisions"])), be.ptr(be.dev(logw)), 10.0, 2e-6, 1e-4, be.ptr(L),
                                  be.ptr(DR), be.ptr(DS), 10, be.ptr(DC), 3, be.ptr(DJ), B, be.ptr(ws), B * 16, be.stream),
          "frame losses")
    Lh = be.host(L)
    e = dict(l2d=abs(Lh[0] - g["l2d"]) / abs(g["l2d"]), lsh=abs(Lh[1] - g["lsh"]) / abs(g["lsh"]),
             lpo=abs(Lh[2] - g["lpo"]) / abs(g["lpo"]), total=abs(Lh[3] - g["loss"]) / abs(g["loss"]),
             drot=rel_err(be.host(DR).reshape(B, 24, 3, 3), gr.numpy()), dshape=rel_err(be.host(DS), gs.numpy()),
             dcam=rel_err(be.host(DC), gc.numpy()), djoints=rel_err(be.host(DJ), gj.numpy()))
    assert max(e.values()) < 5e-4, e
    return e


def case_projection(be, B=3, npnt=49, seed=6):
    rng = _rng(seed)
    cam = torch.tensor([[0.9, 0.02, -0.04]]).repeat(B, 1) + 0.05 * torch.from_numpy(rng.standard_normal((B, 3)).astype(np.float32))
    cam = cam.clone().requires_grad_(True)
    p3 = torch.from_numpy((rng.standard_normal((B, npnt, 3)) * 0.4).astype(np.float32)).requires_grad_(True)
    p2 = O.projection(cam, p3)
    g2 = torch.from_numpy(rng.standard_normal((B, npnt, 2)).astype(np.float32))
    gc, gp = torch.autograd.grad(p2, [cam, p3], g2)
    P2, DP, DC = be.empty((B, npnt, 2)), be.empty((B, npnt, 3)), be.empty((B, 3))
    C_, P3 = be.dev(cam.detach().numpy()), be.dev(p3.detach().numpy())
    check(be.lib.dyb_projection_fwd(be.ptr(C_), 3, be.ptr(P3), be.ptr(P2), B, npnt, be.stream), "proj fwd")
    check(be.lib.dyb_projection_bwd(be.ptr(C_), 3, be.ptr(P3), be.ptr(be.dev(g2.numpy())), be.ptr(DP), be.ptr(DC), 3, B, npnt,
                                    be.stream), "proj bwd")
    e = dict(fwd=rel_err(be.host(P2), p2.detach().numpy()), dp=rel_err(be.host(DP), gp.numpy()), dcam=rel_err(be.host(DC), gc.numpy()))
    assert max(e.values()) < 1e-4, e
    return e


def case_perspective_projection(be, B=3, npnt=49, seed=16):
    """dyb_perspective_projection_* against the reference's formula (utils/geometry.py:63-91) written out in torch."""
    rng = _rng(seed)
    pts = torch.from_numpy((rng.standard_normal((B, npnt, 3)) * 0.4).astype(np.float32)).requires_grad_(True)
    q, _ = np.linalg.qr(rng.standard_normal((B, 3, 3)))
    rot = torch.from_numpy(q.astype(np.float32))
    tr = torch.from_numpy(np.array([[0.05, -0.03, 40.0]], np.float32).repeat(B, 0) + rng.standard_normal((B, 3)).astype(np.float32) * 0.1).requires_grad_(True)
    focal = torch.tensor([5000.0, 4000.0, 1200.0][:B])
    cen = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32) * 10)
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = focal; K[:, 1, 1] = focal; K[:, 2, 2] = 1.0; K[:, :-1, -1] = cen
    p = torch.einsum('bij,bkj->bki', rot, pts) + tr.unsqueeze(1)
    ref = torch.einsum('bij,bkj->bki', K, p / p[:, :, -1].unsqueeze(-1))[:, :, :-1]
    g2 = torch.from_numpy(rng.standard_normal((B, npnt, 2)).astype(np.float32))
    gp, gt = torch.autograd.grad(ref, [pts, tr], g2)
    P, R, T, F_, C = (be.dev(t.detach().numpy()) for t in (pts, rot, tr, focal, cen))
    OUT, DP, DT = be.empty((B, npnt, 2)), be.empty((B, npnt, 3)), be.empty((B, 3))
    check(be.lib.dyb_perspective_projection_fwd(be.ptr(P), be.ptr(R), be.ptr(T), be.ptr(F_), 1, be.ptr(C), be.ptr(OUT), B, npnt, be.stream), "persp fwd")
    check(be.lib.dyb_perspective_projection_bwd(be.ptr(P), be.ptr(R), be.ptr(T), be.ptr(F_), 1, be.ptr(be.dev(g2.numpy())), be.ptr(DP), be.ptr(DT),
                                                B, npnt, be.stream), "persp bwd")
    e = dict(fwd=rel_err(be.host(OUT), ref.detach().numpy()), dp=rel_err(be.host(DP), gp.numpy()), dt=rel_err(be.host(DT), gt.numpy()))
    assert max(e.values()) < 1e-4, e
    return e


def case_gmm_prior(be, gmm, B=5, seed=17):
    """dyb_gmm_prior against MaxMixturePrior.merged_log_likelihood restated in torch (utils/smplify/prior.py:181-196), incl. golden g2."""
    rng = _rng(seed)
    means, prec, w = (torch.from_numpy(np.asarray(gmm[k], np.float32)) for k in ("means", "precisions", "nll_weights"))
    pose = (means[rng.integers(0, 8, B)] + torch.from_numpy(rng.standard_normal((B, 69)).astype(np.float32)) * 0.3).requires_grad_(True)
    d = pose.unsqueeze(1) - means
    ll = 0.5 * (torch.einsum('mij,bmj->bmi', prec, d) * d).sum(-1) - torch.log(w.reshape(1, -1))
    ref = ll.min(1)[0]
    (gp,) = torch.autograd.grad(ref.sum(), [pose])
    logw = np.log(np.asarray(gmm["nll_weights"], np.float32).reshape(-1))
    OUT, DP = be.empty((B,)), be.empty((B, 69))
    check(be.lib.dyb_gmm_prior(be.ptr(be.dev(pose.detach().numpy())), be.ptr(be.dev(means.numpy())), be.ptr(be.dev(prec.numpy())), be.ptr(be.dev(logw)),
                               be.ptr(OUT), be.ptr(DP), B, be.stream), "gmm prior")
    e = dict(val=rel_err(be.host(OUT), ref.detach().numpy()), dpose=rel_err(be.host(DP), gp.numpy()))
    assert max(e.values()) < 1e-4, e
    return e


# ---------------------------------------------------------------------------------------- optimiser family
def case_optim(be, n=4096 * 3, seed=7):
    rng = _rng(seed)
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=3e-6, betas=(0.5, 0.9), foreach=False)
    P, M, V = be.dev(p0), be.zeros((n,)), be.zeros((n,))
    for t in range(1, 4):
        g = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 2)).astype(np.float32)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        check(be.lib.dyb_adam_step(be.ptr(P), be.ptr(be.dev(g)), be.ptr(M), be.ptr(V), 0.5, 0.9, 3e-6 / (1 - 0.5 ** t),
                                   (1 - 0.9 ** t) ** 0.5, 1e-8, n, be.stream), "adam")
    st = opt.state[pt]
    e = dict(adam_delta=rel_err(be.host(P).astype(np.float64) - p0, pt.detach().numpy().astype(np.float64) - p0),
             adam_m=rel_err(be.host(M), st["exp_avg"].numpy()), adam_v=rel_err(be.host(V), st["exp_avg_sq"].numpy()))
    g = rng.standard_normal(n).astype(np.float32)
    OUT = be.empty((n,))
    check(be.lib.dyb_fastweight_update(be.ptr(be.dev(p0)), be.ptr(be.dev(g)), be.ptr(OUT), 8e-6, n, be.stream), "fastweight")
    e["fast"] = rel_err(be.host(OUT).astype(np.float64) - p0, -8e-6 * g.astype(np.float64))
    Tt = be.dev(p0)
    check(be.lib.dyb_ema_update(be.ptr(Tt), be.ptr(be.dev(g)), 0.1, n, be.stream), "ema")
    e["ema"] = rel_err(be.host(Tt), 0.1 * p0 + 0.9 * g)
    Yb = be.dev(p0)
    check(be.lib.dyb_axpby(be.ptr(be.dev(g)), be.ptr(Yb), 0.5, 2.0, n, be.stream), "axpby")
    e["axpby"] = rel_err(be.host(Yb), 0.5 * g + 2.0 * p0)
    a = rng.standard_normal(5000).astype(np.float32)
    b = (a + 0.01 * rng.standard_normal(5000)).astype(np.float32)
    Cs = be.empty((1,))
    check(be.lib.dyb_cosine_sim(be.ptr(be.dev(a)), be.ptr(be.dev(b)), 5000, 1e-12, be.ptr(Cs), be.stream), "cosine")
    ref = float(F.cosine_similarity(torch.from_numpy(a), torch.from_numpy(b), dim=0, eps=1e-12))
    e["cos"] = abs(float(be.host(Cs)[0]) - ref)
    assert e["adam_delta"] < 3e-2 and e["adam_m"] < 1e-5 and e["adam_v"] < 1e-5, e
    assert e["fast"] < 1e-2 and e["ema"] < 1e-6 and e["axpby"] < 1e-6 and e["cos"] < 1e-6, e
    return e


# ---------------------------------------------------------------------------------------- HMR engine
def case_hmr_engine(be, golden, ckpt, check_grads=True):
    """dyb_hmr_forward/backward vs golden g3 = the REFERENCE's own HMR module (model/hmr.py) on
    the same seeded checkpoint and frame: outputs, features, and parameter gradients."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr_layout import HmrLayout
    g = golden("g3_hmr.npz")
    B = 2
    L = HmrLayout(be.lib, B)
    params = be.dev(L.pack(ckpt).numpy())
    img = assets.make_frame(0, batch_size=B, seed=22)["image"].numpy()
    init = np.repeat(HmrLayout.init_state(ckpt).numpy(), B, 0)
    acts = be.empty((L.act_floats,))
    ws = be.empty((L.ws_bytes // 4,))
    check(be.lib.dyb_hmr_forward(L.plan, be.ptr(params), be.ptr(be.dev(img)), be.ptr(be.dev(init)), 3, be.ptr(acts),
                                 be.ptr(ws), L.ws_bytes, be.stream), "hmr forward")
    A = be.host(acts)
    rot = A[L.off_rotmat:L.off_rotmat + B * 216].reshape(B, 24, 3, 3)
    st = A[L.off_state:L.off_state + B * 160].reshape(B, 160)
    e = dict(rotmat=rel_err(rot, g["rotmat"]), shape=rel_err(st[:, 144:154], g["shape"]), cam=rel_err(st[:, 154:157], g["cam"]))

    def feat(i):
        f = L.features[i]
        d = [x for x in f["dims"] if x > 0]
        if len(d) == 4:
            return A[f["offset"]:f["offset"] + int(np.prod(d))].reshape(d)
        rows = A[f["offset"]:f["offset"] + d[0] * f["row_stride"]].reshape(d[0], f["row_stride"])
        return rows[:, :d[1]]
    e["feat5"] = rel_err(feat(5), g["feat5"])
    e["feat12"] = rel_err(feat(12), g["feat12"])
    for i in range(15):
        e[f"abs{i}"] = abs(float(np.abs(feat(i).astype(np.float64)).sum()) - g["feat_abs"][i]) / g["feat_abs"][i]
    assert max(e.values()) < 5e-4, e
    if not check_grads:
        return e
    d_state = np.zeros((B, 160), np.float32)
    d_state[:, 144:154] = g["ws"]
    d_state[:, 154:157] = g["wc"]
    grads = be.zeros((L.n_params,))
    check(be.lib.dyb_hmr_backward(L.plan, be.ptr(params), be.ptr(acts), be.ptr(be.dev(g["wr"])), be.ptr(be.dev(d_state)), 3,
                                  be.ptr(grads), be.ptr(ws), L.ws_bytes, be.stream, be.aux_stream()), "hmr backward")
    G = L.unpack(torch.from_numpy(be.host(grads)))
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([float(G[n].double().norm()) for n in names])
    # 5e-3: a handful of activations sit within fp32 rounding of zero, so their ReLU mask (and with it
    # a 3x3 patch of upstream gradient) legitimately differs between summation orders
    bad = [(n, a, b) for n, a, b in zip(names, norms, g["grad_norms"]) if abs(a - b) > 5e-3 * b]
    assert not bad, bad[:8]
    from conftest import cosine
    for k in g.files:
        if k.startswith("gs_"):
            n = k[3:]
            assert cosine(G[n].flatten()[:256], g[k]) > 0.9999, n
    e["grad_norm_maxrel"] = float(np.abs(norms / g["grad_norms"] - 1).max())
    return e


def case_hmr_engine_schedules(be, ckpt, options, seed=22):
    """One image through the engine (forward + backward) under the default latency schedule and again with `options`
    (dyb_set_option name -> value; restored afterwards) - e.g. the throughput schedule with the one-pass GroupNorm backward, which
    only exists at one image per sequence: every parameter gradient of the second run against the first (itself pinned to the
    reference module's golden g3 at batch 2 by case_hmr_engine)."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr_layout import HmrLayout
    import ctypes
    B = 1
    L = HmrLayout(be.lib, B)
    params = be.dev(L.pack(ckpt).numpy())
    img = assets.make_frame(0, batch_size=B, seed=seed)["image"].numpy()
    init = np.repeat(HmrLayout.init_state(ckpt).numpy(), B, 0)
    rng = _rng(seed)
    d_rot = (rng.standard_normal((B, 24, 3, 3)) * 1e-2).astype(np.float32)
    d_state = np.zeros((B, 160), np.float32)
    d_state[:, 144:157] = (rng.standard_normal((B, 13)) * 1e-2).astype(np.float32)

    def run():
        acts = be.empty((L.act_floats,))
        ws = be.empty((L.ws_bytes // 4,))
        check(be.lib.dyb_hmr_forward(L.plan, be.ptr(params), be.ptr(be.dev(img)), be.ptr(be.dev(init)), 3, be.ptr(acts),
                                     be.ptr(ws), L.ws_bytes, be.stream), "hmr forward")
        grads = be.zeros((L.n_params,))
        check(be.lib.dyb_hmr_backward(L.plan, be.ptr(params), be.ptr(acts), be.ptr(be.dev(d_rot)), be.ptr(be.dev(d_state)), 3,
                                      be.ptr(grads), be.ptr(ws), L.ws_bytes, be.stream, be.aux_stream()), "hmr backward")
        return L.unpack(torch.from_numpy(be.host(grads)))
    ref = run()
    saved = {}
    for k, v in options.items():
        cur = ctypes.c_int()
        be.lib.dyb_get_option(k.encode(), ctypes.byref(cur))
        saved[k] = cur.value
        assert be.lib.dyb_set_option(k.encode(), v) == 0, k
    try:
        got = run()
    finally:
        for k, v in saved.items():
            be.lib.dyb_set_option(k.encode(), v)
    from conftest import cosine
    worst, wcos = 0.0, 1.0
    for n in ref:
        a, b = got[n].double(), ref[n].double()
        worst = max(worst, float((a - b).norm() / b.norm().clamp_min(1e-30)))
        wcos = min(wcos, cosine(a.flatten()[:4096].numpy(), b.flatten()[:4096].numpy()))
    # (the two schedules sum in different orders; activations within rounding of zero may flip their ReLU mask - see case_hmr_engine)
    assert worst < 5e-3 and wcos > 0.9999, (worst, wcos)
    return dict(worst_rel=worst, worst_cos=wcos)


def case_groupnorm_fold(be, N, HW, C, nslabs, with_addend, seed=11):
    """dyb_groupnorm_bwd_fold == dyb_groupnorm_bwd on the pre-folded gradient."""
    rng = _rng(seed)
    y = (rng.standard_normal((N, HW, C)) * 1.3).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(C)).astype(np.float32)
    slabs = rng.standard_normal((nslabs, N, HW, C)).astype(np.float32)
    addend = rng.standard_normal((N, HW, C)).astype(np.float32) if with_addend else None
    dout = slabs.sum(0) + (addend if with_addend else 0)
    wsb = be.lib.dyb_groupnorm_workspace_bytes(N, HW, C)
    ws = be.empty((wsb // 4,))
    Y, OUT, ST = be.dev(y), be.empty((N, HW, C)), be.empty((N, 4, 2))
    G_, B_ = be.dev(gamma), be.dev(beta)
    check(be.lib.dyb_groupnorm_fwd(None, 1, be.ptr(Y), be.ptr(G_), be.ptr(B_), None, be.ptr(OUT), be.ptr(ST), N, HW, C, 1,
                                   be.ptr(ws), wsb, be.stream), "gn fwd")
    res = {}
    for tag in ("ref", "fold"):
        DY, DRES, DG, DB = be.empty((N, HW, C)), be.empty((N, HW, C)), be.empty((C,)), be.empty((C,))
        if tag == "ref":
            check(be.lib.dyb_groupnorm_bwd(be.ptr(be.dev(dout)), be.ptr(OUT), be.ptr(Y), be.ptr(ST), be.ptr(G_), be.ptr(DY),
                                           be.ptr(DRES), be.ptr(DG), be.ptr(DB), N, HW, C, 1, be.ptr(ws), wsb, be.stream), "bwd")
        else:
            FOLD = be.empty((N, HW, C))
            check(be.lib.dyb_groupnorm_bwd_fold(be.ptr(be.dev(slabs)), nslabs, N * HW * C, be.ptr(be.dev(addend)) if with_addend else None,
                                                be.ptr(FOLD), be.ptr(OUT), be.ptr(Y), be.ptr(ST), be.ptr(G_), be.ptr(DY), be.ptr(DRES),
                                                be.ptr(DG), be.ptr(DB), N, HW, C, 1, be.ptr(ws), wsb, be.stream), "bwd fold")
            res["folded"] = rel_err(be.host(FOLD), dout)
        res[tag] = [be.host(a) for a in (DY, DRES, DG, DB)]
    e = dict(folded=res["folded"])
    for name, a, b in zip(("dy", "dres", "dgamma", "dbeta"), res["fold"], res["ref"]):
        e[name] = rel_err(a, b)
    assert max(e.values()) < 1e-5, e
    return e


def case_groupnorm_onepass(be, HW, C, relu, mask_from_y, nslabs, with_addend, cap=0, seed=13):
    """dyb_groupnorm_bwd_onepass (one image; the throughput schedule's GroupNorm backward) against torch autograd of
    relu?(group_norm(y)): dy, dgamma, dbeta and the masked gradient dm.  mask_from_y: the activation was never saved (layers
    inside a bottleneck), the ReLU mask is recomputed from y; cap: float4 per workgroup (small values force several row chunks
    per (image, group) slab, whose workgroups meet on a counter)."""
    rng = _rng(seed)
    y = (rng.standard_normal((1, HW, C)) * 1.4 + 0.2).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(C)).astype(np.float32)
    slabs = rng.standard_normal((nslabs, 1, HW, C)).astype(np.float32)
    addend = rng.standard_normal((1, HW, C)).astype(np.float32) if with_addend else None
    dout = slabs.sum(0) + (addend if with_addend else 0)
    yt = torch.from_numpy(y).permute(0, 2, 1).reshape(1, C, HW, 1).contiguous().requires_grad_(True)
    gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    o = F.group_norm(yt, 4, gt, bt, 1e-5)
    pre = o
    if relu:
        o = F.relu(o)
    gy, gg, gb = torch.autograd.grad(o, [yt, gt, bt], torch.from_numpy(dout).permute(0, 2, 1).reshape(1, C, HW, 1))
    dy_ref = gy.reshape(1, C, HW).permute(0, 2, 1).numpy()
    act = o.detach().reshape(1, C, HW).permute(0, 2, 1).numpy()
    dm_ref = dout * (pre.detach().reshape(1, C, HW).permute(0, 2, 1).numpy() > 0) if relu else dout
    # statistics exactly as the forward saves them
    wsb = be.lib.dyb_groupnorm_workspace_bytes(1, HW, C)
    ws = be.empty((max(wsb, 16) // 4,))
    Y, OUT, ST = be.dev(y), be.empty((1, HW, C)), be.empty((1, 4, 2))
    G_, B_ = be.dev(gamma), be.dev(beta)
    check(be.lib.dyb_groupnorm_fwd(None, 1, be.ptr(Y), be.ptr(G_), be.ptr(B_), None, be.ptr(OUT), be.ptr(ST), 1, HW, C, relu,
                                   be.ptr(ws), wsb, be.stream), "gn fwd")
    assert rel_err(be.host(OUT), act) < 5e-4
    wsb2 = be.lib.dyb_groupnorm_bwd_onepass_workspace_bytes(HW, C)
    ws2 = be.empty((wsb2 // 4,))
    DM, DY, DG, DB = be.empty((1, HW, C)), be.empty((1, HW, C)), be.empty((C,)), be.empty((C,))
    check(be.lib.dyb_groupnorm_bwd_onepass(be.ptr(be.dev(slabs)), nslabs, HW * C, be.ptr(be.dev(addend)) if with_addend else None,
                                           None if (mask_from_y or not relu) else be.ptr(OUT), be.ptr(Y), be.ptr(ST), be.ptr(G_), be.ptr(B_),
                                           be.ptr(DM), be.ptr(DY), be.ptr(DG), be.ptr(DB), HW, C, relu, cap, be.ptr(ws2), wsb2, be.stream),
          "gn bwd onepass")
    e = dict(dy=rel_err(be.host(DY), dy_ref), dm=rel_err(be.host(DM), dm_ref), dgamma=rel_err(be.host(DG), gg.numpy()),
             dbeta=rel_err(be.host(DB), gb.numpy()))
    assert max(e.values()) < 5e-4, e
    return e


# ------------------------------------------------------------------- conv + GroupNorm backward, dy never materialised
def case_conv_gn_bwd_fused(be, N, H, W, C, K, R, stride, pad, relu=1, seed=21):
    """out = relu?(GN(conv(x, w))): gradients w.r.t. x (+ addend), w, gamma, beta through
    dyb_groupnorm_bwd_reduce -> dyb_conv2d_nhwc_dgrad_gn / _wgrad_gn, against torch autograd."""
    rng = _rng(seed)
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((R, R, C, K)) / np.sqrt(R * R * C)).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(K)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(K)).astype(np.float32)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    HW = Ho * Wo
    dout = rng.standard_normal((N, Ho, Wo, K)).astype(np.float32)
    add = rng.standard_normal((N, H, W, C)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wt = torch.from_numpy(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    o = F.group_norm(F.conv2d(xt, wt, stride=stride, padding=pad), 4, gt, bt, 1e-5)
    if relu:
        o = F.relu(o)
    gx, gw, gg, gb = torch.autograd.grad(o, [xt, wt, gt, bt], torch.from_numpy(dout).permute(0, 3, 1, 2))

    wsb = max(be.lib.dyb_conv2d_workspace_bytes(N, H, W, C, K, R, R, stride, pad), 16)
    ws = be.empty((wsb // 4,))
    gwsb = be.lib.dyb_groupnorm_workspace_bytes(N, HW, K)
    gws = be.empty((gwsb // 4,))
    X, Wt, GA, BE_, DO, ADD = be.dev(x), be.dev(w), be.dev(gamma), be.dev(beta), be.dev(dout), be.dev(add)
    Y, OUT, ST = be.empty((N, HW, K)), be.empty((N, HW, K)), be.empty((N, 4, 2))
    check(be.lib.dyb_conv2d_nhwc_fwd(be.ptr(X), be.ptr(Wt), be.ptr(Y), N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb,
                                     be.stream), "conv fwd")
    check(be.lib.dyb_groupnorm_fwd(None, 1, be.ptr(Y), be.ptr(GA), be.ptr(BE_), None, be.ptr(OUT), be.ptr(ST), N, HW, K, relu,
                                   be.ptr(gws), gwsb, be.stream), "gn fwd")
    part = be.empty((be.lib.dyb_groupnorm_bwd_partial_floats(N, HW, K),))
    DM = be.empty((N, HW, K)) if relu else DO
    check(be.lib.dyb_groupnorm_bwd_reduce(be.ptr(DO), be.ptr(OUT), be.ptr(Y), be.ptr(ST), be.ptr(GA), be.ptr(BE_), be.ptr(DM), be.ptr(part),
                                          N, HW, K, relu, be.stream), "gn bwd reduce")
    DX, DW, DG, DB = be.empty(x.shape), be.empty(w.shape), be.empty((K,)), be.empty((K,))
    check(be.lib.dyb_conv2d_nhwc_dgrad_gn(be.ptr(DM), be.ptr(Y), be.ptr(ST), be.ptr(part), be.ptr(GA), be.ptr(Wt), be.ptr(DX),
                                          be.ptr(ADD), N, H, W, C, K, R, R, stride, pad, be.ptr(ws), wsb, be.stream), "dgrad gn")
    ws2 = be.empty((wsb // 4,))
    check(be.lib.dyb_conv2d_nhwc_wgrad_gn(be.ptr(X), be.ptr(DM), be.ptr(Y), be.ptr(ST), be.ptr(part), be.ptr(GA), be.ptr(DW),
                                          be.ptr(DG), be.ptr(DB), N, H, W, C, K, R, R, stride, pad, be.ptr(ws2), wsb,
                                          be.stream), "wgrad gn")
    e = dict(dx=rel_err(be.host(DX), gx.permute(0, 2, 3, 1).numpy() + add),
             dw=rel_err(be.host(DW), gw.permute(2, 3, 1, 0).numpy()),
             dgamma=rel_err(be.host(DG), gg.numpy()), dbeta=rel_err(be.host(DB), gb.numpy()))
    assert max(e.values()) < 5e-4, e
    return e


# ------------------------------------------------------------------- a whole bottleneck through the fused entry points
def case_bottleneck_fused(be, N, H, W, Cin, planes, stride, downsample, seed=31):
    """reference model/hmr.py:40-60 (Bottleneck): conv1-bn1-relu-conv2-bn2-relu-conv3-bn3 (+ downsample) -relu with the
    launch structure of the engine: bn1 / bn2 / downsample.1 outputs never materialised (normalised in the consumer's
    loader), dy never materialised (formed in the dgrad / wgrad loaders).  Output and every gradient vs torch autograd."""
    rng = _rng(seed)
    Cout = planes * 4
    assert downsample or (Cin == Cout and stride == 1)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1

    def wgt(R, Ci, Co):
        return (rng.standard_normal((R, R, Ci, Co)) / np.sqrt(R * R * Ci)).astype(np.float32)

    def gb(C):
        return (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), (0.2 * rng.standard_normal(C)).astype(np.float32)

    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w1, w2, w3 = wgt(1, Cin, planes), wgt(3, planes, planes), wgt(1, planes, Cout)
    (g1, b1), (g2, b2), (g3, b3) = gb(planes), gb(planes), gb(Cout)
    wd = wgt(1, Cin, Cout) if downsample else None
    gd, bd = gb(Cout) if downsample else (None, None)
    dout = rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)

    # ---- torch reference
    T = lambda a: torch.from_numpy(a).requires_grad_(True)
    cw = lambda w: torch.from_numpy(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    tw1, tw2, tw3 = cw(w1), cw(w2), cw(w3)
    tg = [T(a) for a in (g1, b1, g2, b2, g3, b3)]
    a1 = F.relu(F.group_norm(F.conv2d(xt, tw1), 4, tg[0], tg[1], 1e-5))
    a2 = F.relu(F.group_norm(F.conv2d(a1, tw2, stride=stride, padding=1), 4, tg[2], tg[3], 1e-5))
    o = F.group_norm(F.conv2d(a2, tw3), 4, tg[4], tg[5], 1e-5)
    leaves = [xt, tw1, tw2, tw3] + tg
    if downsample:
        twd, tgd, tbd = cw(wd), T(gd), T(bd)
        o = o + F.group_norm(F.conv2d(xt, twd, stride=stride), 4, tgd, tbd, 1e-5)
        leaves += [twd, tgd, tbd]
    else:
        o = o + xt
    # a pre-activation within rounding distance of 0 may take either side of the final ReLU in two correct fp32
    # evaluations, and one flipped element there moves every gradient by O(1e-3): take those elements out of play
    near0 = (o.detach().abs() < 1e-5).permute(0, 2, 3, 1).numpy()
    dout[near0] = 0.0
    o = F.relu(o)
    grads = torch.autograd.grad(o, leaves, torch.from_numpy(dout).permute(0, 3, 1, 2))
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).numpy()
    rscw = lambda t: t.detach().permute(2, 3, 1, 0).numpy()

    # ---- device: forward as the engine issues it
    L = be.lib
    X, DO = be.dev(x), be.dev(dout)
    W1, W2, W3 = be.dev(w1), be.dev(w2), be.dev(w3)
    G1, B1, G2, B2, G3, B3 = [be.dev(a) for a in (g1, b1, g2, b2, g3, b3)]
    shapes = dict(c1=(N, H, W, Cin, planes, 1, 1, 1, 0), c2=(N, H, W, planes, planes, 3, 3, stride, 1),
                  c3=(N, Ho, Wo, planes, Cout, 1, 1, 1, 0), cd=(N, H, W, Cin, Cout, 1, 1, stride, 0))
    wsb = max(max(L.dyb_conv2d_workspace_bytes(*sh) for sh in shapes.values()), 16)
    ws, ws2 = be.empty((wsb // 4,)), be.empty((wsb // 4,))
    HW1, HW2 = H * W, Ho * Wo
    pA = be.empty((max(L.dyb_groupnorm_workspace_bytes(N, HW1, planes), L.dyb_groupnorm_workspace_bytes(N, HW2, Cout)) // 4,))
    pB = be.empty((L.dyb_groupnorm_workspace_bytes(N, HW2, planes) // 4,))
    pD = be.empty((L.dyb_groupnorm_workspace_bytes(N, HW2, Cout) // 4,))
    Y1, Y2, Y3 = be.empty((N, HW1, planes)), be.empty((N, HW2, planes)), be.empty((N, HW2, Cout))
    S1, S2, S3, SD = [be.empty((N, 4, 2)) for _ in range(4)]
    OUT = be.empty((N, HW2, Cout))
    check(L.dyb_conv2d_nhwc_fwd(be.ptr(X), be.ptr(W1), be.ptr(Y1), *shapes["c1"], be.ptr(ws), wsb, be.stream), "c1")
    check(L.dyb_groupnorm_stats(None, 1, be.ptr(Y1), be.ptr(pA), N, HW1, planes, be.stream), "s1")
    check(L.dyb_conv2d_nhwc_fwd_gnin(be.ptr(Y1), be.ptr(pA), be.ptr(G1), be.ptr(B1), 1, be.ptr(S1), be.ptr(W2), be.ptr(Y2),
                                     *shapes["c2"], be.ptr(ws), wsb, be.stream), "c2")
    check(L.dyb_groupnorm_stats(None, 1, be.ptr(Y2), be.ptr(pB), N, HW2, planes, be.stream), "s2")
    if downsample:
        WD, GD, BD, YD = be.dev(wd), be.dev(gd), be.dev(bd), be.empty((N, HW2, Cout))
        check(L.dyb_conv2d_nhwc_fwd(be.ptr(X), be.ptr(WD), be.ptr(YD), *shapes["cd"], be.ptr(ws), wsb, be.stream), "cd")
        check(L.dyb_groupnorm_stats(None, 1, be.ptr(YD), be.ptr(pD), N, HW2, Cout, be.stream), "sd")
    check(L.dyb_conv2d_nhwc_fwd_gnin(be.ptr(Y2), be.ptr(pB), be.ptr(G2), be.ptr(B2), 1, be.ptr(S2), be.ptr(W3), be.ptr(Y3),
                                     *shapes["c3"], be.ptr(ws), wsb, be.stream), "c3")
    check(L.dyb_groupnorm_stats(None, 1, be.ptr(Y3), be.ptr(pA), N, HW2, Cout, be.stream), "s3")
    if downsample:
        check(L.dyb_groupnorm_apply(be.ptr(Y3), be.ptr(pA), be.ptr(G3), be.ptr(B3), be.ptr(YD), be.ptr(pD), be.ptr(GD),
                                    be.ptr(BD), be.ptr(SD), be.ptr(OUT), be.ptr(S3), N, HW2, Cout, 1, be.stream), "a3")
    else:
        check(L.dyb_groupnorm_apply(be.ptr(Y3), be.ptr(pA), be.ptr(G3), be.ptr(B3), be.ptr(X), None, None, None, None,
                                    be.ptr(OUT), be.ptr(S3), N, HW2, Cout, 1, be.stream), "a3")
    e = dict(out=rel_err(be.host(OUT).reshape(N, Ho, Wo, Cout), nhwc(o)))

    # ---- backward as the engine issues it
    def part(HW, C):
        return be.empty((L.dyb_groupnorm_bwd_partial_floats(N, HW, C),))
    P1, P2, P3, PD = part(HW1, planes), part(HW2, planes), part(HW2, Cout), part(HW2, Cout)
    DM3, DM2, DM1 = be.empty((N, HW2, Cout)), be.empty((N, HW2, planes)), be.empty((N, HW1, planes))
    DW1, DW2, DW3 = be.empty(w1.shape), be.empty(w2.shape), be.empty(w3.shape)
    DG = {k: be.empty((c,)) for k, c in dict(g1=planes, b1=planes, g2=planes, b2=planes, g3=Cout, b3=Cout, gd=Cout, bd=Cout).items()}
    T3, T2, DX, RB = be.empty((N, HW2, planes)), be.empty((N, HW1, planes)), be.empty(x.shape), be.empty(x.shape)
    check(L.dyb_groupnorm_bwd_reduce(be.ptr(DO), be.ptr(OUT), be.ptr(Y3), be.ptr(S3), be.ptr(G3), be.ptr(B3), be.ptr(DM3),
                                     be.ptr(P3), N, HW2, Cout, 1, be.stream), "r3")
    check(L.dyb_conv2d_nhwc_wgrad_gn_gnin(be.ptr(Y2), be.ptr(S2), be.ptr(G2), be.ptr(B2), 1, be.ptr(DM3), be.ptr(Y3), be.ptr(S3),
                                          be.ptr(P3), be.ptr(G3), be.ptr(DW3), be.ptr(DG["g3"]), be.ptr(DG["b3"]), *shapes["c3"],
                                          be.ptr(ws2), wsb, be.stream), "wg3")
    check(L.dyb_conv2d_nhwc_dgrad_gn(be.ptr(DM3), be.ptr(Y3), be.ptr(S3), be.ptr(P3), be.ptr(G3), be.ptr(W3), be.ptr(T3), None,
                                     *shapes["c3"], be.ptr(ws), wsb, be.stream), "dg3")
    check(L.dyb_groupnorm_bwd_reduce(be.ptr(T3), None, be.ptr(Y2), be.ptr(S2), be.ptr(G2), be.ptr(B2), be.ptr(DM2), be.ptr(P2),
                                     N, HW2, planes, 1, be.stream), "r2")          # mask recomputed from y2
    check(L.dyb_conv2d_nhwc_wgrad_gn_gnin(be.ptr(Y1), be.ptr(S1), be.ptr(G1), be.ptr(B1), 1, be.ptr(DM2), be.ptr(Y2), be.ptr(S2),
                                          be.ptr(P2), be.ptr(G2), be.ptr(DW2), be.ptr(DG["g2"]), be.ptr(DG["b2"]), *shapes["c2"],
                                          be.ptr(ws2), wsb, be.stream), "wg2")
    check(L.dyb_conv2d_nhwc_dgrad_gn(be.ptr(DM2), be.ptr(Y2), be.ptr(S2), be.ptr(P2), be.ptr(G2), be.ptr(W2), be.ptr(T2), None,
                                     *shapes["c2"], be.ptr(ws), wsb, be.stream), "dg2")
    check(L.dyb_groupnorm_bwd_reduce(be.ptr(T2), None, be.ptr(Y1), be.ptr(S1), be.ptr(G1), be.ptr(B1), be.ptr(DM1), be.ptr(P1),
                                     N, HW1, planes, 1, be.stream), "r1")
    check(L.dyb_conv2d_nhwc_wgrad_gn(be.ptr(X), be.ptr(DM1), be.ptr(Y1), be.ptr(S1), be.ptr(P1), be.ptr(G1), be.ptr(DW1),
                                     be.ptr(DG["g1"]), be.ptr(DG["b1"]), *shapes["c1"], be.ptr(ws2), wsb, be.stream), "wg1")
    if downsample:
        DWD = be.empty(wd.shape)
        check(L.dyb_groupnorm_bwd_reduce(be.ptr(DM3), None, be.ptr(YD), be.ptr(SD), be.ptr(GD), be.ptr(BD), be.ptr(DM3),
                                         be.ptr(PD), N, HW2, Cout, 0, be.stream), "rd")     # no ReLU: dm aliases dout
        check(L.dyb_conv2d_nhwc_wgrad_gn(be.ptr(X), be.ptr(DM3), be.ptr(YD), be.ptr(SD), be.ptr(PD), be.ptr(GD), be.ptr(DWD),
                                         be.ptr(DG["gd"]), be.ptr(DG["bd"]), *shapes["cd"], be.ptr(ws2), wsb, be.stream), "wgd")
        check(L.dyb_conv2d_nhwc_dgrad_gn(be.ptr(DM3), be.ptr(YD), be.ptr(SD), be.ptr(PD), be.ptr(GD), be.ptr(WD), be.ptr(RB),
                                         None, *shapes["cd"], be.ptr(ws), wsb, be.stream), "dgd")
        addend = RB
    else:
        addend = DM3
    check(L.dyb_conv2d_nhwc_dgrad_gn(be.ptr(DM1), be.ptr(Y1), be.ptr(S1), be.ptr(P1), be.ptr(G1), be.ptr(W1), be.ptr(DX),
                                     be.ptr(addend), *shapes["c1"], be.ptr(ws), wsb, be.stream), "dg1")
    e["dx"] = rel_err(be.host(DX), nhwc(grads[0]))
    for i, (k, D) in enumerate([("dw1", DW1), ("dw2", DW2), ("dw3", DW3)]):
        e[k] = rel_err(be.host(D), rscw(grads[1 + i]))
    for i, k in enumerate(["g1", "b1", "g2", "b2", "g3", "b3"]):
        e["d" + k] = rel_err(be.host(DG[k]), grads[4 + i].numpy())
    if downsample:
        e["dwd"] = rel_err(be.host(DWD), rscw(grads[10]))
        e["dgd"] = rel_err(be.host(DG["gd"]), grads[11].numpy())
        e["dbd"] = rel_err(be.host(DG["bd"]), grads[12].numpy())
    assert max(e.values()) < 5e-4, e
    return e


# ------------------------------------------------------------------- PA-MPJPE / Procrustes on the device
def case_pa_mpjpe(be, golden, n_random=40, seed=41):
    """dyb_pa_mpjpe against (a) the reference's compute_similarity_transform_batch outputs (golden g6: random pairs,
    an exact similarity transform, a reflection) and (b) the NumPy restatement on random + degenerate inputs."""
    from dynaboa_amd.pose_utils import compute_similarity_transform_batch
    g = golden("g6_procrustes.npz")
    rng = _rng(seed)
    S1 = rng.normal(0, 0.3, (n_random, 14, 3)).astype(np.float32)
    S2 = (S1 * 1.3 + rng.normal(0, 0.05, S1.shape)).astype(np.float32)
    S2[0] = S1[0]                                   # identical sets: zero error
    S1[1, :, 2] = 0.0; S2[1, :, 2] = 0.0            # coplanar points: rank-2 covariance
    A = np.concatenate([g["S1"].astype(np.float32), S1]); Bm = np.concatenate([g["S2"].astype(np.float32), S2])
    hat_ref = np.concatenate([g["S1_hat"], compute_similarity_transform_batch(S1, S2)])
    err_ref = np.sqrt(((hat_ref - Bm) ** 2).sum(-1)).mean(-1)
    n = A.shape[0]
    out, hat = be.empty((n,)), be.empty((n, 14, 3))
    check(be.lib.dyb_pa_mpjpe(be.ptr(be.dev(A)), be.ptr(be.dev(Bm)), be.ptr(out), be.ptr(hat), n, 14, be.stream), "pa_mpjpe")
    o, h = be.host(out), be.host(hat)
    keep = np.ones(n, bool)
    keep[len(g["S1"]) + 1] = False                  # coplanar case: the aligned points may differ by the free reflection ...
    assert np.abs(h[keep] - hat_ref[keep]).max() < 2e-5 * max(1.0, np.abs(hat_ref).max()), np.abs(h[keep] - hat_ref[keep]).max()
    assert np.abs(o - err_ref).max() < 2e-5 + 1e-4 * err_ref.max(), (o - err_ref)     # ... the error does not
    assert o[len(g["S1"])] < 1e-6
    return dict(max_abs=float(np.abs(o - err_ref).max()))


# ------------------------------------------------------------------- conv + GroupNorm statistics per layer (K4 / tiled)
def case_layer_gnstats(be, N, H, W, C, Ka, Ra, sa, Kb, Rb, sb, seed=51):
    """Two chained layers through dyb_conv2d_nhwc_fwd_gnstats (+ dyb_groupnorm_apply_n at the end):
    out = relu(GN(convB(relu(GN(convA(x)))))) with A's normalised output never materialised.  Shapes pick the
    single-launch K4 kernel (1x1, batch 1, small maps) or the tiled conv + statistics kernel; both must agree with torch."""
    rng = _rng(seed)
    pa, pb = (1 if Ra == 3 else 0), (1 if Rb == 3 else 0)
    Ha, Wa = (H + 2 * pa - Ra) // sa + 1, (W + 2 * pa - Ra) // sa + 1
    Hb, Wb = (Ha + 2 * pb - Rb) // sb + 1, (Wa + 2 * pb - Rb) // sb + 1
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    wa = (rng.standard_normal((Ra, Ra, C, Ka)) / np.sqrt(Ra * Ra * C)).astype(np.float32)
    wb = (rng.standard_normal((Rb, Rb, Ka, Kb)) / np.sqrt(Rb * Rb * Ka)).astype(np.float32)
    ga, ba = (1 + 0.2 * rng.standard_normal(Ka)).astype(np.float32), (0.2 * rng.standard_normal(Ka)).astype(np.float32)
    gb, bb = (1 + 0.2 * rng.standard_normal(Kb)).astype(np.float32), (0.2 * rng.standard_normal(Kb)).astype(np.float32)
    cw = lambda w: torch.from_numpy(w).permute(3, 2, 0, 1).contiguous()
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    ya_t = F.conv2d(xt, cw(wa), stride=sa, padding=pa)
    a_t = F.relu(F.group_norm(ya_t, 4, torch.from_numpy(ga), torch.from_numpy(ba), 1e-5))
    yb_t = F.conv2d(a_t, cw(wb), stride=sb, padding=pb)
    out_t = F.relu(F.group_norm(yb_t, 4, torch.from_numpy(gb), torch.from_numpy(bb), 1e-5))
    L = be.lib
    shA, shB = (N, H, W, C, Ka, Ra, Ra, sa, pa), (N, Ha, Wa, Ka, Kb, Rb, Rb, sb, pb)
    wsb = max(L.dyb_conv2d_workspace_bytes(*shA), L.dyb_conv2d_workspace_bytes(*shB), 16)
    ws = be.empty((wsb // 4,))
    pfl = lambda HW, K: max(L.dyb_groupnorm_workspace_bytes(N, HW, K) // 4, (HW // 32 + 1) * (K // 32 + 1) * 8)
    PA, PB = be.empty((pfl(Ha * Wa, Ka),)), be.empty((pfl(Hb * Wb, Kb),))
    YA, YB = be.empty((N, Ha * Wa, Ka)), be.empty((N, Hb * Wb, Kb))
    SA, SB, OUT = be.empty((N, 4, 2)), be.empty((N, 4, 2)), be.empty((N, Hb * Wb, Kb))
    nA, nB = ctypes.c_int(0), ctypes.c_int(0)
    X, WA, WB, GA, BA, GB_, BB = [be.dev(a) for a in (x, wa, wb, ga, ba, gb, bb)]
    check(L.dyb_conv2d_nhwc_fwd_gnstats(be.ptr(X), None, 0, None, None, 0, None, be.ptr(WA), be.ptr(YA), be.ptr(PA),
                                        ctypes.byref(nA), *shA, be.ptr(ws), wsb, be.stream), "layer A")
    check(L.dyb_conv2d_nhwc_fwd_gnstats(be.ptr(YA), be.ptr(PA), nA.value, be.ptr(GA), be.ptr(BA), 1, be.ptr(SA), be.ptr(WB),
                                        be.ptr(YB), be.ptr(PB), ctypes.byref(nB), *shB, be.ptr(ws), wsb, be.stream), "layer B")
    check(L.dyb_groupnorm_apply_n(be.ptr(YB), be.ptr(PB), nB.value, be.ptr(GB_), be.ptr(BB), None, None, 0, None, None, None,
                                  be.ptr(OUT), be.ptr(SB), N, Hb * Wb, Kb, 1, be.stream), "apply")
    nhwc = lambda t, K: t.permute(0, 2, 3, 1).reshape(N, -1, K).numpy()
    mean_a = ya_t.reshape(N, 4, -1).mean(-1).numpy()
    e = dict(ya=rel_err(be.host(YA), nhwc(ya_t, Ka)), yb=rel_err(be.host(YB), nhwc(yb_t, Kb)), out=rel_err(be.host(OUT), nhwc(out_t, Kb)),
             mean_a=float(np.abs(be.host(SA)[:, :, 0] - mean_a).max()))
    assert max(e.values()) < 5e-4, e
    return dict(e, nA=nA.value, nB=nB.value)


# ------------------------------------------------------------------- data gradient + producer's reduce (K4 backward)
def case_dgrad_gn_reduce(be, H, W, C, Kc, mask_from_y, with_addend, seed=61, N=1):
    """Two chained layers P -> L at batch 1:  a = relu(GN_P(y_p)) [+ residual];  y = conv1x1_L(a);  out = relu(GN_L(y)).
    Given d(out), dyb_conv2d_nhwc_dgrad_gn_reduce must produce P's masked gradient dm_p and a partial block from which P's
    own fused gradients come out right: checked through the folded per-channel sums (= dbeta_P / dgamma_P) and through
    the data gradient of a 1x1 conv placed in front of P.  Runs with DYB_K4_BWD=1 (one launch) and =0 (two launches)."""
    import os
    rng = _rng(seed)
    M = H * W
    y_p = (rng.standard_normal((N, M, C)) * 1.2).astype(np.float32)
    gp, bp = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), (0.2 * rng.standard_normal(C)).astype(np.float32)
    res = rng.standard_normal((N, M, C)).astype(np.float32)           # residual operand of P's activation (bn3 flavour)
    wl = (rng.standard_normal((1, 1, C, Kc)) / np.sqrt(C)).astype(np.float32)
    gl, bl = (1 + 0.2 * rng.standard_normal(Kc)).astype(np.float32), (0.2 * rng.standard_normal(Kc)).astype(np.float32)
    dout = rng.standard_normal((N, M, Kc)).astype(np.float32)
    add = rng.standard_normal((N, M, C)).astype(np.float32) if with_addend else None
    # ---- torch reference
    T = lambda a: torch.from_numpy(a)
    ypt = T(y_p).permute(0, 2, 1).reshape(N, C, H, W).requires_grad_(True)
    gpt, bpt = T(gp).requires_grad_(True), T(bp).requires_grad_(True)
    pre = F.group_norm(ypt, 4, gpt, bpt, 1e-5)
    if not mask_from_y:
        pre = pre + T(res).permute(0, 2, 1).reshape(N, C, H, W)
    a = F.relu(pre)
    yl = F.conv2d(a, T(wl).permute(3, 2, 0, 1))
    ol = F.relu(F.group_norm(yl, 4, T(gl), T(bl), 1e-5))
    extra = (a * T(add).permute(0, 2, 1).reshape(N, C, H, W)).sum() if with_addend else 0.0     # a second consumer of `a`
    loss = (ol * T(dout).permute(0, 2, 1).reshape(N, Kc, H, W)).sum() + extra
    gy, gg, gb = torch.autograd.grad(loss, [ypt, gpt, bpt])
    dyp_ref = gy.reshape(N, C, M).permute(0, 2, 1).numpy()
    L = be.lib
    res_out = {}
    for mode in (1, 0):
        L.dyb_set_option(b"k4_bwd", mode)
        try:
            shL = (N, H, W, C, Kc, 1, 1, 1, 0)
            wsb = max(L.dyb_conv2d_workspace_bytes(*shL), 16)
            ws = be.empty((wsb // 4,))
            gws = be.empty((max(L.dyb_groupnorm_workspace_bytes(N, M, C), L.dyb_groupnorm_workspace_bytes(N, M, Kc)) // 4,))
            gwsb = gws.numel() * 4 if hasattr(gws, "numel") else gws.size * 4
            YP, GP, BP = be.dev(y_p), be.dev(gp), be.dev(bp)
            OUTP, SP = be.empty((N, M, C)), be.empty((N, 4, 2))
            check(L.dyb_groupnorm_fwd(None, 1, be.ptr(YP), be.ptr(GP), be.ptr(BP), None if mask_from_y else be.ptr(be.dev(res)),
                                      be.ptr(OUTP), be.ptr(SP), N, M, C, 1, be.ptr(gws), gwsb, be.stream), "gn P")
            WL, GL, BL = be.dev(wl), be.dev(gl), be.dev(bl)
            YL, OUTL, SL = be.empty((N, M, Kc)), be.empty((N, M, Kc)), be.empty((N, 4, 2))
            check(L.dyb_conv2d_nhwc_fwd(be.ptr(OUTP), be.ptr(WL), be.ptr(YL), *shL, be.ptr(ws), wsb, be.stream), "conv L")
            check(L.dyb_groupnorm_fwd(None, 1, be.ptr(YL), be.ptr(GL), be.ptr(BL), None, be.ptr(OUTL), be.ptr(SL), N, M, Kc, 1,
                                      be.ptr(gws), gwsb, be.stream), "gn L")
            partL = be.empty((L.dyb_groupnorm_bwd_partial_floats(N, M, Kc),))
            DML = be.empty((N, M, Kc))
            check(L.dyb_groupnorm_bwd_reduce(be.ptr(be.dev(dout)), be.ptr(OUTL), be.ptr(YL), be.ptr(SL), be.ptr(GL), be.ptr(BL),
                                             be.ptr(DML), be.ptr(partL), N, M, Kc, 1, be.stream), "reduce L")
            partP = be.empty((L.dyb_groupnorm_bwd_partial_floats(N, M, C) + 64 * C,))
            DMP, DX = be.empty((N, M, C)), be.empty((N, M, C))
            nch, ncolb = ctypes.c_int(0), ctypes.c_int(0)
            check(L.dyb_conv2d_nhwc_dgrad_gn_reduce(be.ptr(DML), be.ptr(YL), be.ptr(SL), be.ptr(partL), 0, 0, be.ptr(GL), be.ptr(WL),
                                                    be.ptr(be.dev(add)) if with_addend else None, be.ptr(YP),
                                                    None if mask_from_y else be.ptr(OUTP), be.ptr(SP), be.ptr(GP), be.ptr(BP),
                                                    be.ptr(DMP), be.ptr(partP), ctypes.byref(nch), ctypes.byref(ncolb), be.ptr(DX),
                                                    *shL, be.ptr(ws), wsb, be.stream), "dgrad+reduce")
            # P's gradients from (dm_p, partials): a 1x1 identity-sized conv "in front of" P gives dy_P as its data gradient
            eye = np.eye(C, dtype=np.float32).reshape(1, 1, C, C)
            DYP, DWI, DGP, DBP = be.empty((N, M, C)), be.empty((1, 1, C, C)), be.empty((C,)), be.empty((C,))
            shI = (N, H, W, C, C, 1, 1, 1, 0)
            wsb2 = max(L.dyb_conv2d_workspace_bytes(*shI), 16)
            ws2 = be.empty((wsb2 // 4,))
            check(L.dyb_conv2d_nhwc_dgrad_gn_n(be.ptr(DMP), be.ptr(YP), be.ptr(SP), be.ptr(partP), nch.value, ncolb.value, be.ptr(GP),
                                               be.ptr(be.dev(eye)), be.ptr(DYP), None, *shI, be.ptr(ws2), wsb2, be.stream), "dgrad P")
            check(L.dyb_conv2d_nhwc_wgrad_gn_n(be.ptr(YP), None, None, None, None, be.ptr(DMP), be.ptr(YP), be.ptr(SP), be.ptr(partP),
                                               nch.value, ncolb.value, be.ptr(GP), be.ptr(DWI), be.ptr(DGP), be.ptr(DBP), *shI,
                                               be.ptr(ws2), wsb2, be.stream), "wgrad P")
            res_out[mode] = dict(dyp=be.host(DYP).copy(), dg=be.host(DGP).copy(), db=be.host(DBP).copy(), dm=be.host(DMP).copy(),
                                 layout=(nch.value, ncolb.value))
        finally:
            L.dyb_set_option(b"k4_bwd", 1)
    e = {}
    for mode, r in res_out.items():
        e[f"dy_p[{mode}]"] = rel_err(r["dyp"], dyp_ref)
        e[f"dgamma_p[{mode}]"] = rel_err(r["dg"], gg.numpy())
        e[f"dbeta_p[{mode}]"] = rel_err(r["db"], gb.numpy())
    e["dm one-vs-two launches"] = rel_err(res_out[1]["dm"], res_out[0]["dm"])
    assert max(e.values()) < 5e-4, e
    return dict(e, layouts=(res_out[1]["layout"], res_out[0]["layout"]))


# ---------------------------------------------------------------------------------------- teacher / motion / labelled terms
def _proj_t(cam, p3):
    tz = 2 * 5000.0 / (224 * cam[:, 0] + 1e-9)
    x, y, z = p3[..., 0] + cam[:, None, 1], p3[..., 1] + cam[:, None, 2], p3[..., 2] + tz[:, None]
    return torch.stack([5000.0 * (x / z) / 112.0, 5000.0 * (y / z) / 112.0], -1)


def case_aux_terms(be, B=3, seed=91):
    """dyb_aux_loss_terms (teacher / motion / labelled-exemplar terms, value + gradient) against torch autograd of the
    reference formulas (base_adaptor.py:320-343, :379-398, :346-376 + :412-422)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    L = be.lib

    def student():
        rot = (torch.eye(3).expand(B, 24, 3, 3) + 0.3 * rn(B, 24, 3, 3)).clone().requires_grad_(True)
        state = torch.zeros(B, 160)
        state[:, 144:154] = rn(B, 10) * 0.5
        state[:, 154:157] = torch.tensor([0.9, 0.02, -0.03]) + 0.05 * rn(B, 3)
        state = state.requires_grad_(True)
        joints = (rn(B, 49, 3) * 0.3).requires_grad_(True)
        return rot, state, joints
    res = {}
    for mode in (0, 1, 2):
        rot, state, joints = student()
        shape, cam = state[:, 144:154], state[:, 154:157]
        s2d = _proj_t(cam, joints)
        w = [0.1, 0.8, 0.1][mode]
        rot2 = torch.eye(3).expand(B, 24, 3, 3) + 0.3 * rn(B, 24, 3, 3)
        st2 = torch.zeros(B, 160)
        st2[:, 144:154] = rn(B, 10) * 0.5
        st2[:, 154:157] = torch.tensor([0.9, 0.02, -0.03]) + 0.05 * rn(B, 3)
        j2 = rn(B, 49, 3) * 0.3
        kp = torch.cat([torch.rand(B, 49, 2, generator=g) * 2 - 1, (torch.rand(B, 49, 1, generator=g) < 0.7).float()], -1)
        kp2 = torch.cat([torch.rand(B, 49, 2, generator=g) * 2 - 1, (torch.rand(B, 49, 1, generator=g) < 0.7).float()], -1)
        gt_rot, gt_betas = torch.eye(3).expand(B, 24, 3, 3) + 0.3 * rn(B, 24, 3, 3), rn(B, 10) * 0.5
        gt_s3d = torch.cat([rn(B, 24, 3) * 0.3, torch.ones(B, 24, 1)], -1)
        extra = []
        if mode == 0:
            t2d = _proj_t(st2[:, 154:157], j2)
            terms = [F.mse_loss(s2d, t2d), F.mse_loss(j2, joints), F.mse_loss(shape, st2[:, 144:154]), F.mse_loss(rot, rot2)]
            loss = terms[0] * 5 + terms[1] * 5 + terms[2] * 0.001 + terms[3]
        elif mode == 1:
            st2 = st2.requires_grad_(True)
            j2 = j2.requires_grad_(True)
            h2d = _proj_t(st2[:, 154:157], j2)
            pm = s2d[:, 25:] - h2d[:, 25:]
            gm = kp[:, 25:, :2] - kp2[:, 25:, :2]
            conf = ((kp2[:, 25:, 2:] + kp[:, 25:, 2:]) == 2).float()
            loss = (((pm - gm) ** 2) * conf).mean()
            terms = [loss, torch.zeros(()), torch.zeros(()), torch.zeros(())]
            extra = [st2, j2]
        else:
            conf = kp[:, 25:, 2:]
            l2d = (((s2d[:, 25:] - kp[:, 25:, :2]) ** 2) * conf).mean()
            p24, g24 = joints[:, 25:], gt_s3d[:, :, :3]
            pc = p24 - ((p24[:, 2] + p24[:, 3]) / 2)[:, None]
            gc = g24 - ((g24[:, 2] + g24[:, 3]) / 2)[:, None]
            l3d = (conf * (pc - gc) ** 2).mean()
            terms = [l2d, l3d, F.mse_loss(shape, gt_betas), F.mse_loss(rot, gt_rot)]
            loss = terms[0] * 5 + terms[1] * 5 + terms[2] * 0.001 + terms[3]
        gs = torch.autograd.grad(loss * w, [rot, state, joints] + extra, allow_unused=True)
        g_rot, g_state, g_j = gs[0], gs[1], gs[2]
        g_rot = torch.zeros_like(rot) if g_rot is None else g_rot
        D = lambda t: be.dev(t.detach().numpy())
        ST, ST2 = D(state), D(st2)
        sp, s2p = be.ptr(ST), be.ptr(ST2)
        ROT, J, ROT2, J2, KP, KP2, GR, GB_, G3 = D(rot), D(joints), D(rot2), D(j2), D(kp), D(kp2), D(gt_rot), D(gt_betas), D(gt_s3d)
        pre = np.float32(0.25)                       # accumulate = 1 adds to what is there
        vals = be.empty((5,))
        outs = {}
        for acc in (0, 1):
            d_rot, d_shape, d_cam, d_j = (be.dev(np.full(s, pre, np.float32)) for s in ((B, 216), (B, 10), (B, 3), (B, 147)))
            d_cam2, d_j2 = be.empty((B, 3)), be.empty((B, 147))
            check(L.dyb_aux_loss_terms(mode, B, acc, w, be.ptr(ROT), sp + 144 * 4, 160, sp + 154 * 4, 160, be.ptr(J), be.ptr(ROT2),
                                       s2p + 144 * 4, 160, s2p + 154 * 4, 160, be.ptr(J2), be.ptr(KP), be.ptr(KP2), be.ptr(GR),
                                       be.ptr(GB_), be.ptr(G3), be.ptr(vals), be.ptr(d_rot), be.ptr(d_shape), be.ptr(d_cam),
                                       be.ptr(d_j), be.ptr(d_cam2), be.ptr(d_j2), be.stream), "aux terms")
            outs[acc] = [be.host(x).copy() for x in (d_rot, d_shape, d_cam, d_j, d_cam2, d_j2)]
        v = be.host(vals)
        e = dict(vals=max(abs(float(v[i]) - float(terms[i])) / (abs(float(terms[i])) + 1e-12) for i in range(4) if float(terms[i]) != 0),
                 total=abs(float(v[4]) - float(loss)) / abs(float(loss)))
        ref = [g_rot.reshape(B, 216).numpy(), g_state[:, 144:154].numpy(), g_state[:, 154:157].numpy(), g_j.reshape(B, 147).numpy()]
        for name, got, want in zip(("d_rot", "d_shape", "d_cam", "d_joints"), outs[0][:4], ref):
            e[name] = float(np.abs(got - want).max() / (np.abs(want).max() + 1e-20)) if np.abs(want).max() > 0 else float(np.abs(got).max())
            e[name + "_acc"] = float(np.abs(outs[1][("d_rot", "d_shape", "d_cam", "d_joints").index(name)] - pre - want).max() /
                                     (np.abs(want).max() + 1e-3))          # (0.25 + g rounds at 3e-8)
        if mode == 1:
            e["d_cam2"] = rel_err(outs[0][4], gs[3][:, 154:157].numpy())
            e["d_joints2"] = rel_err(outs[0][5], gs[4].reshape(B, 147).numpy())
        assert max(e.values()) < 2e-4, (mode, e)
        res[mode] = e
    return res


# ---------------------------------------------------------------------------------------- tangent (JVP) kernels
def case_gn_jvp(be, N, HW, C, relu, with_res, split_ty=False, seed=31, onepass=False):
    """dyb_gn_jvp_fwd / dyb_gn_jvp_bwd against torch (float64): forward tangent of relu?(GN(y) + res) along (ty, tgamma, tbeta,
    tres), and the tangent of its backward (dy, dgamma, dbeta, dres) along the same direction plus tdout.  onepass: the one-launch
    entry points (a slab's chunks meet on an arrival counter inside the launch)."""
    rng = _rng(seed)
    f32 = lambda *s: rng.standard_normal(s).astype(np.float32)
    y, ty = f32(N, HW, C) * 1.5 + 0.3, f32(N, HW, C)
    gamma, beta = (1 + 0.2 * f32(C)), 0.2 * f32(C)
    tgamma, tbeta = f32(C), f32(C)
    res, tres = (f32(N, HW, C), f32(N, HW, C)) if with_res else (None, None)
    dout, tdout = f32(N, HW, C), f32(N, HW, C)
    T = lambda a: torch.from_numpy(a).double()
    to4 = lambda a: a.permute(0, 2, 1).reshape(N, C, HW, 1)
    back = lambda a: a.reshape(N, C, HW).permute(0, 2, 1)

    def fwd(yv, gv, bv, rv):
        o = F.group_norm(to4(yv), 4, gv, bv, 1e-5)
        if rv is not None:
            o = o + to4(rv)
        return back(F.relu(o) if relu else o)

    prim = (T(y), T(gamma), T(beta)) + ((T(res),) if with_res else ())
    tang = (T(ty), T(tgamma), T(tbeta)) + ((T(tres),) if with_res else ())
    f = (lambda a, b, c, d: fwd(a, b, c, d)) if with_res else (lambda a, b, c: fwd(a, b, c, None))
    out_ref, tout_ref = torch.autograd.functional.jvp(f, prim, tang)

    def bwd(*args):            # (primal inputs..., dout) -> gradients w.r.t. the primal inputs
        ins = [a.requires_grad_(True) for a in args[:-1]]
        o = f(*ins)
        return torch.autograd.grad(o, ins, args[-1], create_graph=True)

    g_ref, tg_ref = torch.autograd.functional.jvp(lambda *a: bwd(*a), prim + (T(dout),), tang + (T(tdout),))

    # forward statistics as the engine has them
    yt = T(y).reshape(N, HW, 4, C // 4)
    mean = yt.mean(dim=(1, 3))
    rstd = 1.0 / torch.sqrt(yt.var(dim=(1, 3), unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], -1).float().numpy()
    Y, TY, ST = be.dev(y), be.dev(ty), be.dev(stats)
    GA, BE_, TG, TB = be.dev(gamma), be.dev(beta), be.dev(tgamma), be.dev(tbeta)
    RES, TRES = (be.dev(res), be.dev(tres)) if with_res else (None, None)
    OUT, TOUT, TST = be.empty((N, HW, C)), be.empty((N, HW, C)), be.empty((N, 4, 2))
    SCR = be.empty((int(be.lib.dyb_gn_jvp_scratch_floats(N, HW, C)),))
    TY2 = None
    if split_ty:            # the tangent arrives as two halves (a convolution's conv(tx, w) + conv(x, tw)); the sums launch adds them
        h = f32(N, HW, C)
        TY, TY2 = be.dev(ty - h), be.dev(h)
    nsync = int(be.lib.dyb_gn_jvp_sync_words(N))
    SY1, SY2 = be.zeros((nsync,)), be.zeros((nsync,))          # 32-bit words, zero on entry
    fa = (be.ptr(Y), be.ptr(TY), be.ptr(TY2) if split_ty else None, be.ptr(ST), be.ptr(GA), be.ptr(BE_), be.ptr(TG), be.ptr(TB),
          be.ptr(RES) if with_res else None, be.ptr(TRES) if with_res else None, be.ptr(OUT), be.ptr(TOUT), be.ptr(TST), be.ptr(SCR))
    if onepass:
        check(be.lib.dyb_gn_jvp_fwd_onepass(*fa, be.ptr(SY1), N, HW, C, relu, be.stream), "gn jvp fwd onepass")
    else:
        check(be.lib.dyb_gn_jvp_fwd(*fa, N, HW, C, relu, be.stream), "gn jvp fwd")
    e = dict(out=rel_err(be.host(OUT), out_ref.numpy()), tout=rel_err(be.host(TOUT), tout_ref.numpy()))
    DM, TDM, DY, TDY = be.empty((N, HW, C)), be.empty((N, HW, C)), be.empty((N, HW, C)), be.empty((N, HW, C))
    TDG, TDB = be.empty((C,)), be.empty((C,))
    DO, TDO = be.dev(dout), be.dev(tdout)
    ba = (be.ptr(DO), be.ptr(TDO), be.ptr(OUT), be.ptr(Y), be.ptr(TY), be.ptr(ST), be.ptr(TST), be.ptr(GA), be.ptr(TG), be.ptr(DM),
          be.ptr(TDM), be.ptr(DY), be.ptr(TDY), be.ptr(SCR))
    if onepass:
        check(be.lib.dyb_gn_jvp_bwd_onepass(*ba, be.ptr(SY2), be.ptr(TDG), be.ptr(TDB), N, HW, C, relu, be.stream), "gn jvp bwd onepass")
        assert be.host(SY1).view(np.uint32)[0] == 0 and be.host(SY2).view(np.uint32)[0] == 0          # no wait timed out
    else:
        check(be.lib.dyb_gn_jvp_bwd(*ba, be.ptr(TDG), be.ptr(TDB), N, HW, C, relu, be.stream), "gn jvp bwd")
    e.update(dy=rel_err(be.host(DY), g_ref[0].detach().numpy()), tdy=rel_err(be.host(TDY), tg_ref[0].numpy()),
             tdgamma=rel_err(be.host(TDG), tg_ref[1].numpy()), tdbeta=rel_err(be.host(TDB), tg_ref[2].numpy()))
    if with_res:
        e.update(dres=rel_err(be.host(DM), g_ref[3].detach().numpy()), tdres=rel_err(be.host(TDM), tg_ref[3].numpy()))
    assert max(e.values()) < 2e-4, e
    return e


def case_hmr_hvp(be, ckpt, seed=5, B=1, side=True):
    """dyb_hmr_jvp_forward / dyb_hmr_jvp_backward (exact Hessian-vector product through HMR) against the oracle differentiated
    twice by torch (CPU, float32): scalar s(theta) = <c, state(theta)>, direction v; checks the tangent of the state and
    H v = grad_theta(<grad_theta s, v>) per tensor."""
    from oracle import ref_cpu as O
    from dynaboa_amd import assets
    from dynaboa_amd.hmr_layout import HmrLayout
    from conftest import cosine
    rng = _rng(seed)
    L = HmrLayout(be.lib, B)
    names = [k for k in ckpt if k not in ("init_pose", "init_shape", "init_cam")]
    vdict = {k: torch.from_numpy(rng.standard_normal(tuple(ckpt[k].shape)).astype(np.float32)) * (0.02 if ckpt[k].dim() > 1 else 0.05)
             for k in names}
    for k in ("init_pose", "init_shape", "init_cam"):
        vdict[k] = torch.zeros_like(ckpt[k])
    params, tparams = be.dev(L.pack(ckpt).numpy()), be.dev(L.pack(vdict).numpy())
    img = assets.make_frame(3, batch_size=B, seed=22)["image"]
    init = np.repeat(HmrLayout.init_state(ckpt).numpy(), B, 0)
    c = rng.standard_normal((B, 157)).astype(np.float32)
    c[:, :144] *= 0.3

    # ---- reference: torch forward-over-reverse on the oracle
    P = {k: ckpt[k].clone().requires_grad_(k in names) for k in ckpt}
    ct = torch.from_numpy(c)

    def state_of(Pd):
        pose, shape, cam = O.hmr_forward(Pd, img, return_pose6d=True)
        return torch.cat([pose, shape, cam], 1)
    plist = [P[k] for k in names]
    st = state_of(P)
    g = torch.autograd.grad((st * ct).sum(), plist, create_graph=True)
    gv = sum((a * vdict[k]).sum() for a, k in zip(g, names))
    hv_ref = dict(zip(names, torch.autograd.grad(gv, plist)))
    # tangent of the state, J v, by forward-mode through double backward on a fresh graph
    st2 = state_of(P)
    u = torch.zeros_like(st2, requires_grad=True)
    gu = torch.autograd.grad(st2, plist, u, create_graph=True)
    jv = sum((a * vdict[k]).sum() for a, k in zip(gu, names))
    tstate_ref = torch.autograd.grad(jv, u)[0].numpy()

    # ---- device
    acts, ws = be.empty((L.act_floats,)), be.empty((L.ws_bytes // 4,))
    IMG, INIT = be.dev(img.numpy()), be.dev(init)
    check(be.lib.dyb_hmr_forward(L.plan, be.ptr(params), be.ptr(IMG), be.ptr(INIT), 3, be.ptr(acts), be.ptr(ws), L.ws_bytes, be.stream),
          "hmr forward")
    nd = int(be.lib.dyb_hmr_hvp_dual_floats(L.plan))
    dual = be.empty((nd,))
    # side stream: a real one on the GPU; on the emulator (streams are ignored, launches run in issue order) any non-null handle
    # takes the engine through its side-stream schedule
    aux = (be.aux_stream() or 1) if side else None
    check(be.lib.dyb_hmr_jvp_forward(L.plan, be.ptr(params), be.ptr(tparams), be.ptr(acts), be.ptr(dual), 3, be.ptr(ws), L.ws_bytes,
                                     be.stream, aux), "jvp forward")
    off = int(be.lib.dyb_hmr_hvp_offset_tstate(L.plan))
    tstate = be.host(dual)[off:off + B * 160].reshape(B, 160)[:, :157]
    e = dict(tstate=rel_err(tstate, tstate_ref))
    assert e["tstate"] < 2e-3, e
    d_state = np.zeros((B, 160), np.float32)
    d_state[:, :157] = c
    hv = be.zeros((L.n_params,))
    DS, TDS = be.dev(d_state), be.zeros((B, 160))          # named: the buffers must outlive the call
    check(be.lib.dyb_hmr_jvp_backward(L.plan, be.ptr(params), be.ptr(tparams), be.ptr(acts), be.ptr(dual), be.ptr(DS), be.ptr(TDS), 3,
                                      be.ptr(hv), be.ptr(ws), L.ws_bytes, be.stream, aux), "jvp backward")
    H = L.unpack(torch.from_numpy(be.host(hv)))
    worst = {}
    for k in names:
        a, b = H[k].double().flatten(), hv_ref[k].double().flatten()
        nb = float(b.norm())
        if nb == 0.0:
            continue
        worst[k] = (float((a - b).norm()) / nb, cosine(a.numpy(), b.numpy()))
    bad = {k: v for k, v in worst.items() if v[0] > 2e-3 or v[1] < 0.9999}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:10]
    e["hv_max_rel"] = max(v[0] for v in worst.values())
    e["hv_min_cos"] = min(v[1] for v in worst.values())
    return e


# ---------------------------------------------------------------------------------------- stream ordering (emulator only)
def case_stream_order(be, ckpt, seed=5):
    """The engine's side-stream schedules under the emulator's lazy stream mode (tests/emu: operations queue per stream and run
    when flushed, one stream to completion first and the others only as far as the recorded event waits demand): the first-order
    backward with its weight gradients on the side stream, and the two tangent passes of the exact Hessian-vector product with
    the off-chain halves of every pair there, must give bit-identical results to the in-line run under both drain orders (chain
    first: side-stream work as late as the waits allow; side stream first: as early as they allow).  A missing wait in either
    direction (a result read before it exists, a buffer overwritten while another stream still reads it) changes the numbers."""
    from dynaboa_amd import assets
    from dynaboa_amd.hmr_layout import HmrLayout
    assert be.name == "emu"
    raw = be.raw
    rng = _rng(seed)
    B = 1
    L = HmrLayout(be.lib, B)
    names = [k for k in ckpt if k not in ("init_pose", "init_shape", "init_cam")]
    vdict = {k: torch.from_numpy(rng.standard_normal(tuple(ckpt[k].shape)).astype(np.float32)) * (0.02 if ckpt[k].dim() > 1 else 0.05)
             for k in names}
    for k in ("init_pose", "init_shape", "init_cam"):
        vdict[k] = torch.zeros_like(ckpt[k])
    params, tparams = be.dev(L.pack(ckpt).numpy()), be.dev(L.pack(vdict).numpy())
    IMG = be.dev(assets.make_frame(3, batch_size=B, seed=22)["image"].numpy())
    INIT = be.dev(np.repeat(HmrLayout.init_state(ckpt).numpy(), B, 0))
    acts, ws = be.empty((L.act_floats,)), be.empty((L.ws_bytes // 4,))
    check(be.lib.dyb_hmr_forward(L.plan, be.ptr(params), be.ptr(IMG), be.ptr(INIT), 3, be.ptr(acts), be.ptr(ws), L.ws_bytes, be.stream),
          "hmr forward")
    d_state = np.zeros((B, 160), np.float32)
    d_state[:, :157] = rng.standard_normal((B, 157)).astype(np.float32)
    DS, TDS = be.dev(d_state), be.dev(0.1 * rng.standard_normal((B, 160)).astype(np.float32))
    DROT = be.dev(rng.standard_normal((B, 24, 3, 3)).astype(np.float32))
    nd = int(be.lib.dyb_hmr_hvp_dual_floats(L.plan))
    off = int(be.lib.dyb_hmr_hvp_offset_tstate(L.plan))

    def run(aux, order):
        lazy = order is not None
        ws[...] = np.nan                              # nothing a previous run left in the workspace may stand in for a result not yet produced
        grads, hv, dual = be.zeros((L.n_params,)), be.zeros((L.n_params,)), be.empty((nd,))
        gcopy, hcopy = be.empty((L.n_params,)), be.empty((L.n_params,))
        used = []
        # each call is followed, before the flush, by a consumer on the chain (a copy of its result): the call's own final join is
        # what orders it after the side stream's last weight gradient
        consumer = {0: lambda: be.lib.dyb_scale_add(None, be.ptr(grads), None, be.ptr(gcopy), L.n_params, be.stream),
                    2: lambda: be.lib.dyb_scale_add(None, be.ptr(hv), None, be.ptr(hcopy), L.n_params, be.stream)}
        for ci, call in enumerate((lambda: be.lib.dyb_hmr_backward(L.plan, be.ptr(params), be.ptr(acts), be.ptr(DROT), be.ptr(DS), 3, be.ptr(grads),
                                                     be.ptr(ws), L.ws_bytes, be.stream, aux),
                     lambda: be.lib.dyb_hmr_jvp_forward(L.plan, be.ptr(params), be.ptr(tparams), be.ptr(acts), be.ptr(dual), 3, be.ptr(ws),
                                                        L.ws_bytes, be.stream, aux),
                     lambda: be.lib.dyb_hmr_jvp_backward(L.plan, be.ptr(params), be.ptr(tparams), be.ptr(acts), be.ptr(dual), be.ptr(DS),
                                                         be.ptr(TDS), 3, be.ptr(hv), be.ptr(ws), L.ws_bytes, be.stream, aux))):
            if lazy:
                raw.emu_lazy(1)
            try:
                check(call(), "engine call")
                if ci in consumer:
                    check(consumer[ci](), "consumer")
            finally:
                if lazy:
                    used.append(raw.emu_flush(order))
                    raw.emu_lazy(0)
        assert np.array_equal(be.host(gcopy), be.host(grads)) and np.array_equal(be.host(hcopy), be.host(hv)), "a consumer ran before the join"
        return be.host(grads), be.host(hv), be.host(dual)[off:off + B * 160].copy(), used

    g0, h0, t0, _ = run(None, None)
    assert np.isfinite(g0).all() and np.isfinite(h0).all() and np.abs(h0).max() > 0
    out = {}
    import os
    old = os.environ.get("DYB_HVP_OVERLAP")
    try:
        # 7: everything off-chain on the side stream (the default); 3: dgrad(dy, tw) in line - the (dy, tdy) ring is then the only
        # thing between the chain and the side stream's weight gradients
        for ov in ((7, 3) if os.environ.get("DYB_EMU_FULL") else (7,)):       # the second setting is opt-in (+1 min)
            os.environ["DYB_HVP_OVERLAP"] = str(ov)
            for order in (0, 1):
                g, h, t, used = run(1, order)                 # any non-null handle is a second stream to the emulator
                assert used == [2, 2, 2], used                # both queues held work in every call
                assert np.array_equal(g, g0), ("first-order backward", ov, order, float(np.abs(g - g0).max()))
                assert np.array_equal(t, t0), ("tangent forward", ov, order, float(np.abs(t - t0).max()))
                assert np.array_equal(h, h0), ("tangent backward", ov, order, float(np.abs(h - h0).max()))
                out[(ov, order)] = used
    finally:
        if old is None:
            os.environ.pop("DYB_HVP_OVERLAP", None)
        else:
            os.environ["DYB_HVP_OVERLAP"] = old
    return out
