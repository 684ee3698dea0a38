#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
ARGS="--seqs 16 --steps 16 --warmup 4 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for cfg in "1 1" "1 0" "0 1" "0 0"; do set -- $cfg
  DYB_K4=$1 DYB_K4_BWD=$2 timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=16 k4=$1 k4_bwd=$2', round(d['value'],1), round(d['ms_per_step'],2))"
done
DYB_BENCH_SMOKE_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --seqs 2 --steps 4 --warmup 1 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0 2>&1 | tail -2 | cut -c1-600
