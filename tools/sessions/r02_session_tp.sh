#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "throughput or replica or hmr_engine" 2>&1 | tail -3
ARGS="--steps 16 --warmup 4 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for S in 8 16; do for TP in 99 8; do
  DYB_TP_MIN=$TP timeout 300 python bench.py --seqs $S $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$S tp_min=$TP', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
DYB_TP_MIN=4 timeout 300 python bench.py --seqs 4 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=4 tp_min=4', round(d['value'],1), round(d['ms_per_step'],2))"
timeout 300 python bench.py --seqs 32 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=32', round(d['value'],1), round(d['ms_per_step'],2))"
