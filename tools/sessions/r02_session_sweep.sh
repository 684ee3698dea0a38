#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
ARGS="--steps 10 --warmup 3 --no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
run() { # label, seqs, env...
  local label=$1 seqs=$2; shift 2
  env "$@" timeout 300 python bench.py --seqs $seqs $ARGS 2>gpurun_out/err_$label.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), round(d['ms_per_step'],2))" || tail -3 gpurun_out/err_$label.log
}
run S16_g256 16 DYB_TP_GRID=256
run S16_g384 16 DYB_TP_GRID=384
run S16_g512 16 DYB_TP_GRID=512
run S16_g768 16 DYB_TP_GRID=768
run S16_g512_noxcd 16 DYB_TP_GRID=512 DYB_TP_XCD=0
run S32_g512 32 DYB_TP_GRID=512
run S32_g1024 32 DYB_TP_GRID=1024
run S32_old 32 DYB_TP_KERNEL=0
run S24_g512 24 DYB_TP_GRID=512
run S48_g512 48 DYB_TP_GRID=512
