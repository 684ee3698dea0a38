#!/usr/bin/env python3
"""Per hardware queue of a rocprofv3 kernel trace (middle half of the run): kernels, busy time, and how much of the time two queues are busy
together - to see whether lockstep groups issued on different HIP streams actually share the chip or sit on one hardware queue."""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                         r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    byq = defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    span = (hi - lo) / 1e6
    print(f"window {span:.1f} ms, {len(rows)} kernels, queues {sorted(byq)}")
    for q, rs in sorted(byq.items()):
        busy = sum(e - s for s, e, *_ in rs) / 1e6
        names = defaultdict(int)
        for r in rs:
            names[r[4][:28]] += 1
        top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
        streams = sorted({r[3] for r in rs})
        print(f"  queue {q}: {len(rs)} kernels, busy {busy:.1f} ms ({busy/span:.2f}), streams {streams}, top {top}")
    # time with k queues busy
    ev = []
    for s, e, q, *_ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, lo, defaultdict(int)
    for t, d in ev:
        hist[depth] += t - last
        last = t
        depth += d
    print("  kernels in flight -> share of the window:", {k: round(v / (hi - lo), 3) for k, v in sorted(hist.items())})


if __name__ == "__main__":
    main(sys.argv[1])
