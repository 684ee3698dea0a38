#!/bin/bash
# buffer-addressed throughput kernel: parity, then throughput at 16 / 32 sequences per launch and one phase probe
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "throughput or replica" 2>&1 | tail -4
ARGS="--steps 10 --warmup 3 --no_cpu_baseline --no_sub_records --percentile_frames 0"
timeout 300 python bench.py --seqs 16 $ARGS --conv_table gpurun_out/table_S16_buf.csv 2>gpurun_out/e_buf16.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=16', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'))"
timeout 400 python bench.py --seqs 32 $ARGS --conv_table gpurun_out/table_S32_buf.csv 2>gpurun_out/e_buf32.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=32', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'))"
timeout 200 python bench.py --seqs 16 --steps 4 --warmup 2 --no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline --probe 0,14,256,256,3 2>&1 >/dev/null | grep PROBE | cut -c1-700
