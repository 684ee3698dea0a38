#!/bin/bash
# per-shape conv timing tables (bench.py --conv_table) at 16 and 1 sequences per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
ARGS="--steps 12 --warmup 4 --no_cpu_baseline --no_sub_records --percentile_frames 0"
DYB_TP_MIN=8 timeout 300 python bench.py --seqs 16 $ARGS --conv_table gpurun_out/table_S16_tp.csv 2>gpurun_out/e1.log | tee gpurun_out/b_S16_tp.json | cut -c1-300
DYB_TP_MIN=99 timeout 300 python bench.py --seqs 16 $ARGS --conv_table gpurun_out/table_S16_k4.csv 2>gpurun_out/e2.log | tee gpurun_out/b_S16_k4.json | cut -c1-300
timeout 300 python bench.py --seqs 1 $ARGS --conv_table gpurun_out/table_S1.csv 2>gpurun_out/e3.log | tee gpurun_out/b_S1.json | cut -c1-300
