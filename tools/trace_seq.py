#!/usr/bin/env python3
"""Kernel-by-kernel timeline of ONE steady-state engine forward and ONE backward out of a rocprofv3
kernel trace of tools/enginebench.py: duration of every kernel and the idle gap in front of it."""
import csv
import sys


def main(path, out):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                         (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])),
                         r.get("Queue_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("nchw3_to_nhwc4")]
    lines = []
    # a forward in the middle of the forward loop
    i0 = starts[len(starts) // 2]
    i1 = starts[len(starts) // 2 + 1]
    seq = rows[i0:i1]
    tot = seq[-1][1] - seq[0][0]
    busy = sum(e - s for s, e, *_ in seq)
    lines.append(f"FORWARD: {len(seq)} kernels span {tot/1e3:.1f} us busy {busy/1e3:.1f} us")
    prev = seq[0][0]
    for s, e, k, g, q in seq:
        lines.append(f"  gap {max(0,s-prev)/1e3:6.2f}  dur {(e-s)/1e3:6.2f}  {k} {g} q{q}")
        prev = e
    # a backward: window between two rot6d_bwd kernels
    bst = [i for i, r in enumerate(rows) if r[2].startswith("rot6d_bwd")]
    if len(bst) > 4:
        j0, j1 = bst[len(bst) // 4], bst[len(bst) // 4 + 1]
        seq = rows[j0:j1]
        tot = seq[-1][1] - seq[0][0]
        lines.append(f"BACKWARD: {len(seq)} kernels span {tot/1e3:.1f} us")
        byq = {}
        for s, e, k, g, q in seq:
            p = byq.get(q, seq[0][0])
            lines.append(f"  t {(s-seq[0][0])/1e3:8.2f} gap {max(0,s-p)/1e3:6.2f}  dur {(e-s)/1e3:6.2f}  {k} {g} q{q}")
            byq[q] = e
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:3]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
