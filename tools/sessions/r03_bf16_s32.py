#!/usr/bin/env python3
"""32 sequences in lockstep with the convolutions on the bf16 matrix cores (bench.py record bf16_S32 alone)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
r = bench.sub_record(dev, "bf16_S", 10, 3, 1, 3, f"{S} sequences, bf16 MFMA", roofline_peak=bench.PEAK_BF16_MFMA_TFLOPS, seqs=S, bf16_mfma=1)
__import__("dynaboa_amd.hmr", fromlist=["get_layout"]).get_layout(1).set_bf16(False)
print(json.dumps({k: v for k, v in r.items() if k != "config"}))
