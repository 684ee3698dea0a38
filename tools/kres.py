#!/usr/bin/env python3
"""Register / scratch / occupancy figures of the kernels of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kres.py igemm_conv.hip [name-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "dynaboa_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = b.split()[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat not in d:
        continue
    f = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-70s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s LDS %s" % (d[:70], f("VGPRs"), f("AGPRs"), f("SGPRs"), f(r"ScratchSize \[bytes/lane\]"),
                                                                         f(r"Occupancy \[waves/SIMD\]"), f(r"LDS Size \[bytes/block\]")))
