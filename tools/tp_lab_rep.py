#!/usr/bin/env python3
"""Kernel lab, replica form: the three convolution modes of ONE layer for S sequence replicas in one launch (own weights each),
timed alone with HIP events, over a list of switch settings.  rep_split = 1 (throughput schedule) unless the spec says otherwise.
    python tools/tp_lab_rep.py S H C K R stride "tp_grid=512" "tp_grid=768,tp_occ=3" ..."""
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
    S, H, C, K, R, s = (int(x) for x in sys.argv[1:7])
    p, W = R // 2, H
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    Ho = (H + 2 * p - R) // s + 1
    x = torch.randn(S, 1, H, W, C, device=dev)
    w = torch.randn(S, R, R, C, K, device=dev) * 0.05
    dy = torch.randn(S, 1, Ho, Ho, K, device=dev)
    outs = [torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)]
    wsb = 1 << 30
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    gflop = 2.0 * S * Ho * Ho * K * R * R * C / 1e9
    defaults = dict(rep_split=1, tp_kernel=2, tp_grid=512, tp_occ=0, tp_xcd=1)
    for spec in sys.argv[7:] or [""]:
        cfg = dict(defaults)
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            cfg[k] = int(v)
        for k, v in cfg.items():
            assert lib.dyb_set_option(k.encode(), v) == 0, k
        row = dict(cfg=spec, shape=[S, H, C, K, R, s])
        for mode, name in enumerate(("fwd", "dgrad", "wgrad")):
            fn = lambda: lib.dyb_debug_conv_replicas(mode, x.data_ptr(), w.data_ptr(), dy.data_ptr(), outs[mode].data_ptr(), S, 1, H, W, C, K, R, R, s, p,
                                                     ws.data_ptr(), wsb, st)
            assert fn() == 0
            us = timeit(fn)
            row[name] = [round(us, 1), round(gflop / us * 1e3, 1)]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
