#!/bin/bash
# lockstep groups on separate streams / host threads: does overlapping one group's streaming phases with another's convolutions pay?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
ARGS="--steps 10 --warmup 3 --no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
run() { timeout 300 python bench.py --seqs $1 --groups $2 $ARGS 2>gpurun_out/e_grp_$1_$2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$1 groups=$2', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('lockstep_groups'), d['config'].get('pa_mpjpe_mm_synthetic_mean'))" || tail -5 gpurun_out/e_grp_$1_$2.log; }
run 4 2
run 32 2
run 32 4
run 32 1
