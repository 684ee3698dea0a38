#!/usr/bin/env python3
"""Host-issue time vs GPU time of whole-engine calls (is the frame host- or GPU-limited?)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynaboa_amd import _lib, assets                      # noqa: E402
from dynaboa_amd.hmr import aux_stream_of, get_layout      # noqa: E402
from dynaboa_amd.hmr_layout import HmrLayout               # noqa: E402


def main(B=1, reps=20):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    L = get_layout(B)
    sd = assets.make_synthetic_checkpoint(22, prefix="")["model"]
    theta = L.pack(sd).to(dev)
    img = torch.randn(B, 3, 224, 224, device=dev)
    init = HmrLayout.init_state(sd).to(dev).repeat(B, 1)
    acts = torch.empty(L.act_floats, device=dev)
    ws = torch.empty(L.ws_bytes, dtype=torch.uint8, device=dev)
    grads = torch.zeros(L.n_params, device=dev)
    d_rot = torch.randn(B, 24, 3, 3, device=dev)
    d_st = torch.randn(B, 160, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    aux = aux_stream_of(theta)

    def fwd():
        lib.dyb_hmr_forward(L.plan, theta.data_ptr(), img.data_ptr(), init.data_ptr(), 3, acts.data_ptr(), ws.data_ptr(), L.ws_bytes, st)

    def bwd(a):
        lib.dyb_hmr_backward(L.plan, theta.data_ptr(), acts.data_ptr(), d_rot.data_ptr(), d_st.data_ptr(), 3, grads.data_ptr(),
                             ws.data_ptr(), L.ws_bytes, st, a)
    out = {}
    for name, fn in (("forward", fwd), ("backward_aux", lambda: bwd(aux)), ("backward_1stream", lambda: bwd(None))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / reps * 1e3
        e1.synchronize()
        out[name] = dict(host_issue_ms=host, gpu_ms=e0.elapsed_time(e1) / reps)
    scratch = torch.zeros(64, device=dev)
    for blocks in (1, 256):
        lib.dyb_debug_launch_chain(scratch.data_ptr(), 50, blocks, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        lib.dyb_debug_launch_chain(scratch.data_ptr(), 2000, blocks, st)
        e1.record()
        host = (time.perf_counter() - t0) * 1e3
        e1.synchronize()
        out[f"chain_{blocks}wg"] = dict(host_us_per_launch=host, gpu_us_per_launch=e0.elapsed_time(e1) / 2.0)
    # graph replay of the same forward / backward (what a captured frame would cost)
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            sst = s.cuda_stream
            g = torch.cuda.CUDAGraph()
            lib.dyb_hmr_forward(L.plan, theta.data_ptr(), img.data_ptr(), init.data_ptr(), 3, acts.data_ptr(), ws.data_ptr(), L.ws_bytes, sst)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                lib.dyb_hmr_forward(L.plan, theta.data_ptr(), img.data_ptr(), init.data_ptr(), 3, acts.data_ptr(), ws.data_ptr(), L.ws_bytes, s.cuda_stream)
                lib.dyb_hmr_backward(L.plan, theta.data_ptr(), acts.data_ptr(), d_rot.data_ptr(), d_st.data_ptr(), 3, grads.data_ptr(),
                                     ws.data_ptr(), L.ws_bytes, s.cuda_stream, None)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(reps):
                g.replay()
            e1.record(s)
            e1.synchronize()
            out["graph_fwd_plus_bwd"] = dict(gpu_ms=e0.elapsed_time(e1) / reps)
    except Exception as e:      # noqa: BLE001
        out["graph_fwd_plus_bwd"] = dict(error=repr(e)[:300])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "enginebench.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
