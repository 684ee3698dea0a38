#!/bin/bash
# throughput-form conv kernel (igemm_tp_kernel): parity tests, then A/B against the 64x64 kernel at 16 sequences per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "throughput or replica" 2>&1 | tail -4
ARGS="--steps 12 --warmup 4 --no_cpu_baseline --no_sub_records --percentile_frames 0"
for TPK in 1 0; do
  DYB_TP_KERNEL=$TPK timeout 300 python bench.py --seqs 16 $ARGS --conv_table gpurun_out/table_S16_tpk$TPK.csv 2>gpurun_out/e_tpk$TPK.log | tee gpurun_out/b_S16_tpk$TPK.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=16 tp_kernel=$TPK', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'))"
done
DYB_TP_GRID=2048 timeout 300 python bench.py --seqs 16 $ARGS --no_roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=16 tp_grid=2048', round(d['value'],1), round(d['ms_per_step'],2))"
DYB_TP_GRID=512 timeout 300 python bench.py --seqs 16 $ARGS --no_roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=16 tp_grid=512', round(d['value'],1), round(d['ms_per_step'],2))"
