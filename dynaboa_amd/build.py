"""Build libdynaboa_hip.so in-tree for gfx950:  python -m dynaboa_amd.build [--force]"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdynaboa_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force: bool = False, verbose: bool = False) -> str:
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    deps = sources() + headers
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in sources():
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
