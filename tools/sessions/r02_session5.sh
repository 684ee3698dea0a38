#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--steps 30 --warmup 6 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for S in 8 16; do for RS in 0 1; do
  DYB_REP_SPLIT=$RS timeout 300 python bench.py --seqs $S $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$S rep_split=$RS', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr1 -o trace -- python $R/bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0) > gpurun_out/tr1.log 2>&1
python tools/frame_timeline.py $(find gpurun_out/tr1 -name "*kernel_trace.csv" | head -1) gpurun_out/frame_timeline_S1.txt
rm -rf gpurun_out/tr1
