#!/usr/bin/env python3
"""Kernel lab for the LATENCY schedule (one image, one sequence): every ResNet-50 conv shape through the plain C ABI - forward, data
gradient, weight gradient (each call includes its split-K fold launch when the policy splits) - timed with HIP events over back-to-back
calls on one stream, i.e. what a dependent chain pays per call.  Prints us per call, the call's algorithmic GFLOP and the effective TFLOP/s,
sorted by the time the shape costs a frame (calls per frame x us).  Diagnostic only.
    python tools/lat_lab.py [out.txt]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynaboa_amd import _lib     # noqa: E402

# (H, W, C, K, R, stride, pad, layers of that shape in the network)
SHAPES = [(224, 224, 4, 64, 7, 2, 3, 1), (56, 56, 64, 64, 1, 1, 0, 1), (56, 56, 64, 64, 3, 1, 1, 3), (56, 56, 64, 256, 1, 1, 0, 4),
          (56, 56, 256, 64, 1, 1, 0, 2), (56, 56, 256, 128, 1, 1, 0, 1), (56, 56, 128, 128, 3, 2, 1, 1), (28, 28, 128, 512, 1, 1, 0, 4),
          (56, 56, 256, 512, 1, 2, 0, 1), (28, 28, 512, 128, 1, 1, 0, 3), (28, 28, 128, 128, 3, 1, 1, 3), (28, 28, 512, 256, 1, 1, 0, 1),
          (28, 28, 256, 256, 3, 2, 1, 1), (14, 14, 256, 1024, 1, 1, 0, 6), (28, 28, 512, 1024, 1, 2, 0, 1), (14, 14, 1024, 256, 1, 1, 0, 5),
          (14, 14, 256, 256, 3, 1, 1, 5), (14, 14, 1024, 512, 1, 1, 0, 1), (14, 14, 512, 512, 3, 2, 1, 1), (7, 7, 512, 2048, 1, 1, 0, 3),
          (14, 14, 1024, 2048, 1, 2, 0, 1), (7, 7, 2048, 512, 1, 1, 0, 2), (7, 7, 512, 512, 3, 1, 1, 2)]


def timeit(fn, reps=40, warm=6):
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
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = []
    for H, W, C, K, R, s, p, nl in SHAPES:
        Ho = (H + 2 * p - R) // s + 1
        x = torch.randn(1, H, W, C, device=dev)
        w = torch.randn(R, R, C, K, device=dev) * 0.05
        dy = torch.randn(1, Ho, Ho, K, device=dev)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
        wsb = max(int(lib.dyb_conv2d_workspace_bytes(1, H, W, C, K, R, R, s, p)), 1 << 24)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        gflop = 2.0 * Ho * Ho * K * R * R * (3 if C == 4 else C) / 1e9
        t = {}
        for name, fn in (("fwd", lambda: lib.dyb_conv2d_nhwc_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st)),
                         ("dgrad", lambda: lib.dyb_conv2d_nhwc_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), None, 1, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st)),
                         ("wgrad", lambda: lib.dyb_conv2d_nhwc_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, H, W, C, K, R, R, s, p, ws.data_ptr(), wsb, st))):
            t[name] = timeit(fn)
        rows.append(((H, C, K, R, s), nl, gflop, t))
    # per frame on the faithful schedule: 5 forwards, 4 data gradients (on the chain), 4 weight gradients (auxiliary stream)
    rows.sort(key=lambda r: -r[1] * (5 * r[3]["fwd"] + 4 * r[3]["dgrad"]))
    L = ["shape (H, C, K, R, stride)   layers  GFLOP   fwd us (TF)    dgrad us (TF)   wgrad us (TF)   chain us per frame (5 fwd + 4 dgrad) x layers"]
    tot = 0.0
    for (sh, nl, g, t) in rows:
        chain = nl * (5 * t["fwd"] + 4 * t["dgrad"])
        tot += chain
        L.append(f"{str(sh):28s} {nl:3d}  {g:6.3f}  " + "  ".join(f"{t[k]:6.1f} ({g / t[k] * 1e3:5.1f})" for k in ("fwd", "dgrad", "wgrad")) + f"   {chain:8.0f}")
    L.append(f"chain total {tot / 1e3:.2f} ms per frame of plain conv calls (the engine's fused-loader / single-launch 1x1 forms are shorter)")
    out = "\n".join(L)
    print(out)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out + "\n")


if __name__ == "__main__":
    main()
