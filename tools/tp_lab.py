#!/usr/bin/env python3
"""Kernel lab for igemm_tp_kernel: ONE convolution (forward / data gradient / weight gradient) at batch N with shared weights through
the plain C ABI in the throughput schedule (tp_batch_min = 1), timed with HIP events, over a list of switch settings.
    python tools/tp_lab.py N H C K R stride  "tp_kernel=2,tp_grid=256,tp_occ=1" "tp_kernel=2,tp_grid=512,tp_occ=2" ...
Prints us per launch and TFLOP/s (algorithmic flops).  Diagnostic only."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynaboa_amd import _lib     # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps      # us


def main():
    N, H, C, K, R, s = (int(x) for x in sys.argv[1:7])
    p = R // 2
    W = H
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    Ho = (H + 2 * p - R) // s + 1
    x = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(R, R, C, K, device=dev) * 0.05
    dy = torch.randn(N, Ho, Ho, K, device=dev)
    y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
    wsb = max(int(lib.dyb_conv2d_workspace_bytes(N, H, W, C, K, R, R, s, p)), 1 << 28)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    gflop = 2.0 * N * Ho * Ho * K * R * R * C / 1e9
    lib.dyb_set_option(b"tp_batch_min", 1)
    defaults = dict(tp_kernel=2, tp_grid=512, tp_occ=0, tp_xcd=1)
    for spec in sys.argv[7:] or [""]:
        cfg = dict(defaults)
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            cfg[k] = int(v)
        for k, v in cfg.items():
            assert lib.dyb_set_option(k.encode(), v) == 0, k
        row = dict(cfg=spec, shape=[N, H, C, K, R, s])
        for name, fn in (("fwd", lambda: lib.dyb_conv2d_nhwc_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st)),
                         ("dgrad", lambda: lib.dyb_conv2d_nhwc_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), None, N, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st)),
                         ("wgrad", lambda: lib.dyb_conv2d_nhwc_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st))):
            us = timeit(fn)
            row[name + "_us"] = round(us, 1)
            row[name + "_tf"] = round(gflop / us * 1e3, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
