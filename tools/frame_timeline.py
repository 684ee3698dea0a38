#!/usr/bin/env python3
"""One steady-state frame step out of a rocprofv3 kernel trace of bench.py (window between two consecutive adam_kernel
launches): per hardware queue the number of kernels, busy time and idle gaps; the critical (main) queue's largest gaps; and
the neighbourhood of every runtime copy / fill kernel (to find what still issues them)."""
import csv
import sys
from collections import defaultdict


def main(path, out):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                         (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])),
                         r.get("Queue_Id", "?")))
    rows.sort()
    # a step ends with Adam: one launch, or - replica groups, ranged updates - three launches within a few hundred microseconds (the first
    # of a burst marks the boundary)
    ad_all = [i for i, r in enumerate(rows) if r[2].startswith(("adam_kernel", "adam_segs_kernel"))]      # (round 6: the segment-list form)
    ad = [i for n, i in enumerate(ad_all) if n == 0 or rows[i][0] - rows[ad_all[n - 1]][0] > 5_000_000]
    i0, i1 = ad[len(ad) // 2], ad[len(ad) // 2 + 1]
    seq = rows[i0:i1]
    span = seq[-1][1] - seq[0][0]
    L = [f"frame step: {len(seq)} kernels, span {span/1e3:.1f} us, sum of kernel time {sum(e-s for s,e,*_ in seq)/1e3:.1f} us"]
    byq = defaultdict(list)
    for r in seq:
        byq[r[4]].append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, *_ in rs)
        gaps = [max(0, rs[i][0] - rs[i - 1][1]) for i in range(1, len(rs))]
        L.append(f"queue {q}: {len(rs)} kernels, busy {busy/1e3:.1f} us, gaps {sum(gaps)/1e3:.1f} us (mean {sum(gaps)/max(1,len(gaps))/1e3:.2f}), "
                 f"from {(rs[0][0]-seq[0][0])/1e3:.0f} to {(rs[-1][1]-seq[0][0])/1e3:.0f} us")
    rs = byq[main_q]
    big = sorted(((rs[i][0] - rs[i - 1][1], i) for i in range(1, len(rs))), reverse=True)[:25]
    L.append("largest gaps on the main queue (gap us: previous kernel -> next kernel):")
    for g, i in big:
        L.append(f"  {g/1e3:7.2f}  {rs[i-1][2]} {rs[i-1][3]} ({(rs[i-1][1]-rs[i-1][0])/1e3:.1f} us) -> {rs[i][2]} {rs[i][3]}")
    L.append("runtime copy / fill kernels and their neighbours (same queue):")
    for q, rs in byq.items():
        for i, r in enumerate(rs):
            if "rocclr" in r[2] or "at::native" in r[2]:
                a = rs[i - 1][2] if i else "-"
                b = rs[i + 1][2] if i + 1 < len(rs) else "-"
                L.append(f"  q{q} t={(r[0]-seq[0][0])/1e3:8.1f} {r[2][:40]} {r[3]} dur {(r[1]-r[0])/1e3:.1f}   after {a[:34]}   before {b[:34]}")
    by = defaultdict(lambda: [0, 0])
    for s, e, k, g, q in seq:
        by[k][0] += 1
        by[k][1] += e - s
    L.append("per kernel in this step:")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
        L.append(f"  {c:5d} x {t/c/1e3:7.2f} us = {t/1e3:8.1f} us  {k}")
    open(out, "w").write("\n".join(L) + "\n")
    print("\n".join(L[:8]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
