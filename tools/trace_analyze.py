#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV on the GPU box: busy fraction of the steady-state window,
gap histogram, and per (kernel, grid) average durations.  Writes a small JSON next to the trace."""
import csv
import json
import sys
from collections import defaultdict

import numpy as np


def main(path, out):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])),
                         int(r["Workgroup_Size_X"]), r.get("Queue_Id", "?")))
    rows.sort()
    n = len(rows)
    lo, hi = int(n * 0.45), int(n * 0.95)            # steady-state window (skip warm-up / tail)
    win = rows[lo:hi]
    span = win[-1][1] - win[0][0]
    busy = sum(e - s for s, e, *_ in win)
    gaps = np.array([max(0, win[i + 1][0] - win[i][1]) for i in range(len(win) - 1)], dtype=np.float64)
    agg = defaultdict(lambda: [0, 0.0])
    byname = defaultdict(lambda: [0, 0.0])
    byq = defaultdict(lambda: [0, 0.0])
    byqk = defaultdict(lambda: [0, 0.0])
    for s, e, name, grid, wg, q in rows[lo:hi]:
        short = name.split("(")[0].replace("void ", "").split("<")[0]
        byname[short][0] += 1
        byname[short][1] += e - s
        byq[q][0] += 1
        byq[q][1] += e - s
        byqk[(q, short)][0] += 1
        byqk[(q, short)][1] += e - s
    for s, e, name, grid, wg, q in rows[lo:hi]:
        short = name.split("(")[0].replace("void ", "")
        blocks = tuple(g // wg if i == 0 else g for i, g in enumerate(grid))
        k = f"{short} {blocks}"
        agg[k][0] += 1
        agg[k][1] += e - s
    table = sorted(((k, c, t / c / 1e3, t / 1e3) for k, (c, t) in agg.items()), key=lambda x: -x[3])
    res = dict(kernels_in_window=len(win), span_ms=span / 1e6, busy_ms=busy / 1e6, busy_frac=busy / span,
               gap_us=dict(mean=float(gaps.mean() / 1e3), p50=float(np.percentile(gaps, 50) / 1e3),
                           p90=float(np.percentile(gaps, 90) / 1e3), p99=float(np.percentile(gaps, 99) / 1e3)),
               by_kernel_grid=[dict(k=k, calls=c, avg_us=round(a, 2), total_us=round(t, 1)) for k, c, a, t in table[:70]])
    res["by_kernel"] = [dict(k=k, calls=c, total_us=round(t / 1e3, 1)) for k, (c, t) in sorted(byname.items(), key=lambda x: -x[1][1])]
    res["by_queue_kernel"] = [dict(q=q, k=k, calls=c, total_us=round(t / 1e3, 1))
                              for (q, k), (c, t) in sorted(byqk.items(), key=lambda x: (x[0][0], -x[1][1]))]
    res["by_queue"] = {q: dict(kernels=c, busy_ms=t / 1e6) for q, (c, t) in byq.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("by_kernel_grid", "by_kernel", "by_queue", "by_queue_kernel")}))
    print(res["by_queue"])
    for r in res["by_kernel"][:40]:
        print(r)
    for r in res["by_queue_kernel"]:
        print(r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
