#!/bin/bash
# r06 s19: soak - default term set, 32 sequences, dynamic loop entered (threshold 1e-4), 40 frames, EMA inside Adam on / off: same PA-MPJPE mean;
# headline at 64 sequences over 100 frames (no hand-off time-outs: the metric flush raises on any)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for fe in 1 0; do
DYB_FUSE_EMA=$fe GPU_MAX_HW_QUEUES=8 timeout 900 python bench.py --full_losses 1 --seqs_full 1 --seqs 32 --inner_step 1 --cos_sim_threshold 1e-4 --steps 40 --warmup 3 --no_cpu_baseline --no_sub_records --no_roofline --percentile_frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('full S32 dynamic fuse_ema=$fe', round(d['value'],1), round(d['ms_per_step'],1), repr(d['config']['pa_mpjpe_mm_synthetic_mean']))"
done
timeout 900 python bench.py --seqs 64 --steps 100 --warmup 5 --no_cpu_baseline --no_sub_records --no_roofline --percentile_frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('S=64, 100 frames', round(d['value'],1), round(d['ms_per_step'],2), repr(d['config']['pa_mpjpe_mm_synthetic_mean']))"
