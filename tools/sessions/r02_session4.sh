#!/bin/bash
# A/B of the single-sequence chain before / after the replica-aware kernels (tree of the previous commit under _ab_prev/),
# then kernel-trace stats at S = 1 and S = 8.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--steps 60 --warmup 10 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for i in 1 2; do
  (cd _ab_prev && timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'], d['host_issue_ms_per_step'])")
  timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('head', d['value'], d['ms_per_step'], d['host_issue_ms_per_step'])"
done
for S in 1 8; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_S$S -o trace -- python $R/bench.py --seqs $S --steps 8 --warmup 2 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0) > gpurun_out/prof_S$S.log 2>&1
  cp $(find gpurun_out/prof_S$S -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_S$S.csv 2>/dev/null
  rm -rf gpurun_out/prof_S$S
  tail -2 gpurun_out/prof_S$S.log | cut -c1-300
done
