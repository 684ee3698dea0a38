#!/usr/bin/env python3
"""Static check for serialised loads: compile every csrc/*.hip to gfx950 assembly and, per kernel, count
"load -> wait for everything" pairs (a global_load whose next vector-memory event is `s_waitcnt vmcnt(0)` with no other
load in between).  On a latency chain each such pair is one exposed ~1.5 us cold-L2 round trip; a healthy kernel issues
its loads in batches (many loads, then counted waits).  Usage: python tools/isa_scan.py [kernel-name-substring]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernels(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s", line)
        if m and ".amdhsa" not in line:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield name, body
                name = None


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:      # noqa: BLE001
        return n


def scan(body):
    loads = pairs = batches = 0
    run = 0
    for line in body:
        t = line.strip()
        if t.startswith("global_load") or t.startswith("buffer_load"):
            loads += 1
            run += 1
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            if run == 1:
                pairs += 1
            if run > 1:
                batches += 1
            run = 0
        elif t.startswith("s_waitcnt") and "vmcnt(" in t:
            if run > 1:
                batches += 1
            run = 0
    return loads, pairs, batches


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for src in sorted(glob.glob(os.path.join(ROOT, "dynaboa_amd", "csrc", "*.hip"))):
            out = os.path.join(d, os.path.basename(src) + ".s")
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out],
                           check=True, stderr=subprocess.DEVNULL)
            for name, body in kernels(open(out).read()):
                dn = demangle(name)
                if flt in dn:
                    rows.append((os.path.basename(src), dn) + scan(body))
    print("%-18s %-52s %6s %12s %8s" % ("file", "kernel", "loads", "load+wait0", "batches"))
    for r in sorted(rows, key=lambda r: -r[3]):
        print("%-18s %-52s %6d %12d %8d" % (r[0], r[1][:52], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
