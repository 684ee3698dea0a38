#!/usr/bin/env python3
"""Where the main stream's time goes inside a frame: HIP events around the phases of Adaptor.adaptation
(level forward+loss, adapt = backward + fast-weight step, outer backward, Adam, inference issue)."""
import os
import sys
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynaboa_amd import assets, benchmark as DB          # noqa: E402
from dynaboa_amd.base_adaptor import synthetic_bundle     # noqa: E402
from dynaboa_amd.maml import MAML                         # noqa: E402

dev = torch.device("cuda:0")
o = DB.frame_only_options(inner_step=3)
o.deferred_metrics = 1
o.overlap_metrics = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True), device=dev)
N, W = 30, 6
frames = [{k: v.to(dev) for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(N + W)]
ad.reset_records(N + W)
spans = []          # (name, ev0, ev1, host_ms)


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        r = fn(*a, **k)
        e1.record()
        spans.append((label, e0, e1, (time.perf_counter() - t0) * 1e3))
        return r
    setattr(obj, name, w)


wrap(ad, "_level", "level fwd+loss")
wrap(MAML, "adapt", "adapt (bwd + fast weights)")
wrap(ad.optimizer, "step", "adam step")
wrap(ad, "inference", "inference (issue)")
wrap(ad, "_join_side", "join side")
orig_backward = torch.Tensor.backward


def bw(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    r = orig_backward(self, *a, **k)
    e1.record()
    spans.append(("outer backward", e0, e1, (time.perf_counter() - t0) * 1e3))
    return r


torch.Tensor.backward = bw
main = torch.cuda.Stream(device=dev)
with torch.cuda.stream(main):
    for s in range(W):
        ad.global_step = s; ad.fit_losses = {}; ad.model.eval(); ad.adaptation(frames[s])
    torch.cuda.synchronize()
    spans.clear()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    f0.record()
    for s in range(W, W + N):
        ad.global_step = s; ad.fit_losses = {}; ad.model.eval(); ad.adaptation(frames[s])
    f1.record()
    t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
agg = defaultdict(lambda: [0, 0.0, 0.0])
for label, e0, e1, h in spans:
    a = agg[label]
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += h
tot = f0.elapsed_time(f1)
print("frames %d  wall %.2f ms/frame  host issue %.2f ms/frame  main-stream span %.2f ms/frame" % (N, wall * 1e3 / N, t_issue * 1e3 / N, tot / N))
acc = 0.0
for label, (c, g, h) in agg.items():
    print("  %-28s calls/frame %.1f   gpu %.3f ms/frame (%.3f each)   host %.3f ms/frame" % (label, c / N, g / N, g / c, h / N))
    acc += g
print("  sum of spans %.2f ms/frame; unaccounted (between spans) %.2f" % (acc / N, (tot - acc) / N))
