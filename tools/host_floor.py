#!/usr/bin/env python3
"""Host cost of ISSUING one natively stepped frame (one sequence, frame-loss set) when the queue is empty: synchronise, issue one
frame without waiting, read the host clock, synchronise again (that frame's device time).  If issuing takes as long as executing,
the single-stream rate is bound by the host's launch rate, not by the device chain.   python tools/host_floor.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch     # noqa: E402
from dynaboa_amd import assets, benchmark as DB     # noqa: E402
from dynaboa_amd.base_adaptor import synthetic_bundle     # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
o = DB.frame_only_options(inner_step=3)
o.deferred_metrics = 1
o.overlap_metrics = 2
ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True), device=dev)
frames = [{k: v.to(dev) for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(N + 10)]
ad.reset_records(N + 10)
st = torch.cuda.Stream()
iss, dev_t = [], []
with torch.cuda.stream(st):
    for s in range(10):
        ad.global_step = s; ad.fit_losses = {}; ad.adaptation(frames[s])
    torch.cuda.synchronize()
    for s in range(10, N + 10):
        t0 = time.perf_counter()
        ad.global_step = s; ad.fit_losses = {}; ad.adaptation(frames[s])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        iss.append((t1 - t0) * 1e3); dev_t.append((t2 - t0) * 1e3)
iss.sort(); dev_t.sort()
print("one frame, queue empty at the start: host issue p50 %.2f ms (min %.2f), issue + drain p50 %.2f ms (min %.2f)" %
      (iss[len(iss) // 2], iss[0], dev_t[len(dev_t) // 2], dev_t[0]))
