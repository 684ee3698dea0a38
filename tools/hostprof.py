"""Where the host time of a natively stepped frame goes (S = 1, frame-loss set): the stepper's own section timers
(dyb_stepper_get_f) against the wall clock of the Python loop around it."""
import sys, time, ctypes
sys.path.insert(0, '/root/repo')
import torch
from dynaboa_amd import assets, benchmark as DB
from dynaboa_amd.base_adaptor import synthetic_bundle
dev = torch.device("cuda:0")
for overlap in (0, 1):
    o = DB.frame_only_options(inner_step=3); o.deferred_metrics = 1; o.overlap_metrics = overlap
    ad = DB.Adaptor(o, synthetic_bundle(seed=22, identity_pose=True), device=dev)
    N = 120
    frames = [{k: v.to(dev) for k, v in assets.make_frame(s, 1, seed=22).items()} for s in range(N)]
    ad.reset_records(N)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for s in range(20):
            ad.global_step = s; ad.fit_losses = {}; ad.adaptation(frames[s])
        torch.cuda.synchronize()
        lib, h = ad._native.lib, ad._native.h
        lib.dyb_stepper_get_f.restype = ctypes.c_double
        base = {k: lib.dyb_stepper_get_f(h, k.encode()) for k in ("host_ms_forward", "host_ms_backward", "host_ms_head", "host_ms_update", "host_ms_tail", "host_ms_total", "host_frames")}
        t0 = time.perf_counter()
        for s in range(20, N):
            ad.global_step = s; ad.fit_losses = {}; ad.adaptation(frames[s])
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    now = {k: lib.dyb_stepper_get_f(h, k.encode()) for k in base}
    n = now["host_frames"] - base["host_frames"]
    print("overlap", overlap, "frames", n, "wall ms/frame %.2f" % (t_all * 1e3 / n), "python loop issue ms/frame %.2f" % (t_issue * 1e3 / n),
          {k: round((now[k] - base[k]) / n, 3) for k in base if k != "host_frames"})
