#!/usr/bin/env python3
"""bench.py's batch-16 records alone (configs[4] arms): python tools/r03_b16.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dynaboa_amd import _lib
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
out = {}
out["fp32"] = bench.sub_record(dev, "b16_fp32", 10, 3, 16, 3, "b16 fp32 throughput schedule", roofline_peak=bench.PEAK_FP32_MFMA_TFLOPS)
out["bf16"] = bench.sub_record(dev, "b16_bf16", 10, 3, 16, 3, "b16 bf16 throughput kernel", roofline_peak=bench.PEAK_BF16_MFMA_TFLOPS, bf16_mfma=1)
__import__("dynaboa_amd.hmr", fromlist=["get_layout"]).get_layout(16).set_bf16(False)
out["so"] = bench.sub_record(dev, "b16_so", 6, 2, 16, 3, "b16 second order exact hvp", second_order=1, hvp="exact")
_lib.load().dyb_set_option(b"tp_batch_min", 0)
out["fp32_lat"] = bench.sub_record(dev, "b16_fp32_lat", 10, 3, 16, 3, "b16 fp32 latency schedule", roofline_peak=bench.PEAK_FP32_MFMA_TFLOPS)
for k, v in out.items():
    print(k, json.dumps({kk: vv for kk, vv in v.items() if kk != "config"}))
