#!/usr/bin/env python3
"""Kernel lab, replica form, many shapes in one process: the three convolution modes of each layer shape for S sequence replicas in one
launch (own weights each), timed alone with HIP events, for every switch setting given.
    python tools/tp_lab_multi.py S "H,C,K,R,stride;H,C,K,R,stride;..." "tp_wt=0" "tp_wt=1" ...
One JSON line per (shape, setting): us per launch and TFLOP/s (algorithmic) of fwd / dgrad / wgrad.  Diagnostic only."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    S = int(sys.argv[1])
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in sys.argv[2].split(";") if sh]
    specs = sys.argv[3:] or [""]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    wsb = 1 << 30
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    defaults = dict(rep_split=1, tp_kernel=2, tp_grid=512, tp_occ=0, tp_xcd=1)
    tot = {}
    for (H, C, K, R, s) in shapes:
        p, W = R // 2, H
        Ho = (H + 2 * p - R) // s + 1
        x = torch.randn(S, 1, H, W, C, device=dev)
        w = torch.randn(S, R, R, C, K, device=dev) * 0.05
        dy = torch.randn(S, 1, Ho, Ho, K, device=dev)
        outs = [torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)]
        gflop = 2.0 * S * Ho * Ho * K * R * R * C / 1e9
        for spec in specs:
            cfg = dict(defaults)
            for kv in filter(None, spec.split(",")):
                k, v = kv.split("=")
                cfg[k] = int(v)
            for k, v in cfg.items():
                assert lib.dyb_set_option(k.encode(), v) == 0, k
            row = dict(shape=[H, C, K, R, s], cfg=spec)
            for mode, name in enumerate(("fwd", "dgrad", "wgrad")):
                fn = lambda: lib.dyb_debug_conv_replicas(mode, x.data_ptr(), w.data_ptr(), dy.data_ptr(), outs[mode].data_ptr(), S, 1, H, W, C, K,
                                                         R, R, s, p, ws.data_ptr(), wsb, st)
                assert fn() == 0
                us = timeit(fn)
                row[name] = [round(us, 1), round(gflop / us * 1e3, 1)]
                t = tot.setdefault(spec, dict(fwd=0.0, dgrad=0.0, wgrad=0.0))
                t[name] += us
            print(json.dumps(row), flush=True)
        del x, w, dy, outs
    for spec, t in tot.items():
        print(json.dumps(dict(total_us=spec, **{k: round(v, 1) for k, v in t.items()}, all=round(sum(t.values()), 1))), flush=True)


if __name__ == "__main__":
    main()
