#!/bin/bash
# r06 s18: where the latency and the throughput schedule cross with round 6's kernels (the fused updates favour the throughput form)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for s in 2 3 4 5 6; do for tm in 1 9; do
DYB_TP_MIN_SEQUENCES=$tm timeout 300 python bench.py --steps 20 --warmup 5 --seqs $s --no_cpu_baseline --no_sub_records --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('S=$s', 'throughput' if $tm == 1 else 'latency   ', round(d['value'],1))"
done; done
