#!/usr/bin/env python3
"""The reference's default flags at S sequences per GPU (bench.py's full_default_losses_S32 record alone): python tools/r03_full_s32.py [S]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
r = bench.sub_record(dev, "full_default_losses_S", 10, 3, 1, 1, f"default flags, {S} sequences in lockstep", roofline_peak=bench.PEAK_FP32_MFMA_TFLOPS,
                     seqs=S, full_losses=1)
print(json.dumps(r))
