#!/usr/bin/env python
"""Print the per-shape conv timing table bench.py --conv_table wrote: TFLOP/s per (kind, shape), sorted by time."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["ms_total"]) for r in rows)
by_kind = {}
for r in rows:
    k = by_kind.setdefault(r["kind"], [0.0, 0.0])
    k[0] += float(r["ms_total"]); k[1] += float(r["gflop_total"])
print("kind  ms      share  TFLOP/s")
for k, (ms, gf) in sorted(by_kind.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:4s} {ms:8.2f} {ms/tot:6.1%} {gf/ms:7.1f}")
print(f"all  {tot:8.2f}        {sum(v[1] for v in by_kind.values())/tot:7.1f}\n")
print("kind N   H   W    C    K R s p split rep launches  us/launch  share  TFLOP/s")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in sorted(rows, key=lambda r: -float(r["ms_total"]))[:top]:
    ms, n, gf = float(r["ms_total"]), int(r["launches"]), float(r["gflop_total"])
    print(f"{r['kind']:4s} {r['N']:>2s} {r['H']:>3s} {r['W']:>3s} {r['C']:>4s} {r['K']:>4s} {r['R']} {r['stride']} {r['pad']} {r['nsplit']:>4s} {r['nrep']:>3s} {n:7d} {ms*1e3/n:9.1f} {ms/tot:6.1%} {gf/ms:7.1f}")
