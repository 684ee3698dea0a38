#!/usr/bin/env python3
"""Per-layer timings of the C-ABI ops at the real ResNet-50 / SMPL shapes (HIP events on torch's
current stream).  Writes gpurun_out/microbench.json; used to steer kernel work, not a headline."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import RESNET_CONVS  # noqa: E402
from dynaboa_amd import _lib     # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps      # us


def main(batch=1):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = []
    for cnt, H, W, C, K, R, s, p in RESNET_CONVS:
        Ho = (H + 2 * p - R) // s + 1
        x = torch.randn(batch, H, W, C, device=dev)
        w = torch.randn(R, R, C, K, device=dev) * 0.05
        dy = torch.randn(batch, Ho, Ho, K, device=dev)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
        out, res = torch.empty_like(dy), torch.randn_like(dy)
        gam, bet = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        stats = torch.empty(batch * 8, device=dev)
        dgam, dbet, dres, dyc = torch.empty(K, device=dev), torch.empty(K, device=dev), torch.empty_like(dy), torch.empty_like(dy)
        wsb = int(lib.dyb_conv2d_workspace_bytes(batch, H, W, C, K, R, R, s, p))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        gwsb = int(lib.dyb_groupnorm_workspace_bytes(batch, Ho * Ho, K))
        gws = torch.empty(gwsb, dtype=torch.uint8, device=dev)
        t = dict(shape=[H, W, C, K, R, s], count=cnt, gflop=2.0 * batch * Ho * Ho * K * R * R * (3 if C == 4 else C) / 1e9)
        t["fwd_us"] = timeit(lambda: lib.dyb_conv2d_nhwc_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), batch, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st))
        if C != 4:
            t["dgrad_us"] = timeit(lambda: lib.dyb_conv2d_nhwc_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), None, batch, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st))
        t["wgrad_us"] = timeit(lambda: lib.dyb_conv2d_nhwc_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), batch, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st))
        t["gn_fwd_us"] = timeit(lambda: lib.dyb_groupnorm_fwd(None, 1, y.data_ptr(), gam.data_ptr(), bet.data_ptr(), res.data_ptr(), out.data_ptr(), stats.data_ptr(), batch, Ho * Ho, K, 1, gws.data_ptr(), gwsb, st))
        t["gn_bwd_us"] = timeit(lambda: lib.dyb_groupnorm_bwd(dy.data_ptr(), out.data_ptr(), y.data_ptr(), stats.data_ptr(), gam.data_ptr(), dyc.data_ptr(), dres.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), batch, Ho * Ho, K, 1, gws.data_ptr(), gwsb, st))
        for k in ("fwd_us", "dgrad_us", "wgrad_us"):
            if k in t:
                t[k.replace("_us", "_tflops")] = t["gflop"] / t[k] * 1e-3 * 1e6 / 1e3
        rows.append(t)
        print(json.dumps(t))
    # empty-kernel launch floor for reference
    a = torch.zeros(4, device=dev)
    floor = timeit(lambda: lib.dyb_axpby(a.data_ptr(), a.data_ptr(), 1.0, 0.0, 4, st), reps=200)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tot = {k: sum(r.get(k, 0) * r["count"] for r in rows) for k in ("fwd_us", "dgrad_us", "wgrad_us", "gn_fwd_us", "gn_bwd_us")}
    json.dump(dict(batch=batch, launch_floor_us=floor, totals_per_pass_us=tot, layers=rows),
              open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w"), indent=1)
    print("floor", floor, "totals", tot)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
