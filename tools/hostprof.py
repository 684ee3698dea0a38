#!/usr/bin/env python3
"""cProfile of the host side of the bench loop (where does the ~22 ms/frame of issue time go?)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynaboa_amd import assets, benchmark as DB          # noqa: E402
from dynaboa_amd.base_adaptor import synthetic_bundle     # noqa: E402

dev = torch.device("cuda:0")
o = DB.frame_only_options(inner_step=3)
o.deferred_metrics = 1
o.overlap_metrics = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True), device=dev)
N = 30
frames = [{k: v.to(dev) for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(N + 5)]
ad.reset_records(N + 5)


def run(lo, hi):
    for s in range(lo, hi):
        ad.global_step = s
        ad.fit_losses = {}
        ad.model.eval()
        ad.adaptation(frames[s])


run(0, 5)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run(5, N + 5)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
import io
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
print(buf.getvalue()[:6000])
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(30)
print(buf.getvalue()[:6000])
