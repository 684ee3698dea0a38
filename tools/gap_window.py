#!/usr/bin/env python3
"""Around the largest idle gap of the main queue inside one steady-state frame step of a rocprofv3 kernel trace: what every queue
ran in the window [gap start - before us, gap end + after us].
    python tools/gap_window.py kernel_trace.csv [before_us] [after_us] [rank]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    before = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
    after = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
    rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48],
                         (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])),
                         r.get("Queue_Id", "?")))
    rows.sort()
    ad_all = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    ad = [i for n, i in enumerate(ad_all) if n == 0 or rows[i][0] - rows[ad_all[n - 1]][0] > 5_000_000]
    i0, i1 = ad[len(ad) // 2], ad[len(ad) // 2 + 1]
    seq = rows[i0:i1]
    byq = defaultdict(list)
    for r in seq:
        byq[r[4]].append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    mq = byq[main_q]
    gaps = sorted(((mq[i + 1][0] - mq[i][1], i) for i in range(len(mq) - 1)), reverse=True)
    g, i = gaps[rank]
    t0, t1 = mq[i][1], mq[i + 1][0]
    base = seq[0][0]
    print(f"gap #{rank} on main queue {main_q}: {g/1e3:.1f} us, from t={(t0-base)/1e3:.1f} to {(t1-base)/1e3:.1f} us of a {(seq[-1][1]-base)/1e3:.0f} us step")
    lo, hi = t0 - before * 1e3, t1 + after * 1e3
    for r in seq:
        if r[1] >= lo and r[0] <= hi:
            print(f"q{r[4]:>2} {(r[0]-base)/1e3:9.1f} .. {(r[1]-base)/1e3:9.1f} ({(r[1]-r[0])/1e3:6.1f}) {r[2]} {r[3]}")


if __name__ == "__main__":
    main()
