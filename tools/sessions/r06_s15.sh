#!/bin/bash
# r06 s15: 200-frame streams of 32 sequences with the fused updates on / off: the synthetic-stream PA-MPJPE means must be the same number
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for f in 1 0; do
DYB_FUSE_FAST=$f DYB_FUSE_ADAM=$f DYB_FUSE_LINEAR=$f timeout 600 python bench.py --steps 200 --warmup 5 --no_cpu_baseline --no_sub_records --no_roofline --percentile_frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fuse=$f', round(d['value'],1), round(d['ms_per_step'],2), repr(d['config']['pa_mpjpe_mm_synthetic_mean']), d['frame_time_ms']['p50'], d['frame_time_ms']['p99'])"
done
