#!/usr/bin/env python3
"""Kernel time of a frame step by family from a rocprofv3 --stats CSV of `bench.py --seqs 32 --steps N` (python tools/step_breakdown.py
<kernel_stats.csv> <steps incl. warm-up>): ms per step, share of kernel time, launches per step."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
groups = {}
for r in rows:
    n = r['Name']
    if 'igemm_tp_kernel<0' in n: g = 'conv forward (igemm_tp)'
    elif 'igemm_tp_kernel<1' in n: g = 'conv data gradient (igemm_tp)'
    elif 'igemm_tp_kernel<2' in n: g = 'conv weight gradient (igemm_tp, auxiliary queue)'
    elif 'igemm' in n: g = 'conv, latency-form kernels (stem forward)'
    elif 'gn_bwd' in n: g = 'GroupNorm backward (reduce + apply)'
    elif 'gn_' in n[:14]: g = 'GroupNorm forward (stats + apply)'
    elif 'fastweight' in n or 'adam' in n: g = 'fast-weight steps + Adam (streaming)'
    elif 'linear' in n: g = 'regressor linears'
    elif 'splitk' in n or 'fold_scatter' in n: g = 'split-K folds'
    elif 'rocclr' in n or 'at::' in n: g = 'runtime copies / fills (setup, flush)'
    elif 'lbs' in n or 'pool' in n: g = 'SMPL LBS + pools'
    else: g = 'other (loss head, metrics, gather, ...)'
    groups.setdefault(g, [0.0, 0]); groups[g][0] += float(r['TotalDurationNs']); groups[g][1] += int(r['Calls'])
print("| kernel family | ms per step | share of kernel time | launches per step |\n|---|---|---|---|")
for g, (t, c) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print(f"| {g} | {t/1e6/steps:.2f} | {t/tot:.1%} | {c/steps:.0f} |")
print(f"| all | {tot/1e6/steps:.1f} | | |")
