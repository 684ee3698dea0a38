#!/bin/bash
# phase-class data gradient (stride 2) in the throughput kernel: parity, then 16 / 32 sequences per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "throughput or replica" 2>&1 | tail -4
ARGS="--steps 10 --warmup 3 --no_cpu_baseline --no_sub_records --percentile_frames 0"
timeout 300 python bench.py --seqs 16 $ARGS --conv_table gpurun_out/table_S16_cls.csv 2>gpurun_out/e_cls16.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=16', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'))"
timeout 400 python bench.py --seqs 32 $ARGS --conv_table gpurun_out/table_S32_cls.csv 2>gpurun_out/e_cls32.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=32', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'))"
