#!/usr/bin/env python3
"""Kernel lab for the one-pass GroupNorm backward: ONE layer for S sequence replicas in one launch, timed alone with HIP events over a
list of switch settings; prints us per launch and GB/s of algorithmic traffic.
    python tools/gn_lab.py S HW C relu mode  "tp_gn_threads=256,tp_gn_poll=1" "tp_gn_threads=256,tp_gn_poll=16" ...
mode bits: 1 = mask from the saved activation (one more tensor read), 2 = write dm (one more tensor written).  (The "tp_gn_poll=-1" rows of
profiles/r04_gn_lab.txt - no wait at all, results wrong, timing of the streaming part - came from a lab-only flag that has been removed
again.)"""
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
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    S, HW, C, relu, mode = (int(x) for x in sys.argv[1:6])
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    n = HW * C
    wsf = int(lib.dyb_groupnorm_bwd_onepass_workspace_bytes(HW, C)) // 4
    per = (5 * n + 8 + 2 * C + wsf + 63) // 64 * 64
    blob = torch.randn(S, per, device=dev)
    blob[:, 5 * n:5 * n + 8] = torch.tensor([0.1, 1.2] * 4, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    passes = 3 + (1 if (relu and mode & 1) else 0) + (1 if mode & 2 else 0)
    gb = passes * n * 4 * S / 1e9
    for spec in sys.argv[6:] or [""]:
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            assert lib.dyb_set_option(k.encode(), int(v)) == 0, k
        fn = lambda: lib.dyb_debug_gn_onepass_replicas(blob.data_ptr(), per, S, gamma.data_ptr(), beta.data_ptr(), HW, C, relu, mode, st)
        assert fn() == 0
        us = timeit(fn)
        print(json.dumps(dict(cfg=spec, shape=[S, HW, C, relu, mode], us=round(us, 1), GBps=round(gb / us * 1e6, 0), passes=passes)), flush=True)


if __name__ == "__main__":
    main()
