#!/bin/bash
# phase clocks of the throughput conv kernel on four representative layers, 16 sequences per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
ARGS="--seqs 16 --steps 4 --warmup 2 --no_cpu_baseline --no_sub_records --percentile_frames 0 --no_roofline"
for P in 0,14,256,256,3 1,14,256,256,3 2,14,256,1024,1 0,56,64,64,3; do
  timeout 200 python bench.py $ARGS --probe $P --probe_out gpurun_out/probe_$P.json 2>&1 >/dev/null | grep PROBE | cut -c1-900
done
