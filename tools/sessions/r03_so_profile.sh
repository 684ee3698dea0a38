#!/bin/bash
# kernel trace of the second-order frame (one sequence, exact Hessian-vector products): where its 51 ms go
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trso -o trace -- python $R/bench.py --seqs 1 --second_order 1 --hvp exact --steps 4 --warmup 2 $Q) > gpurun_out/so_trace.log 2>&1
f=$(find gpurun_out/trso -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/so_kernel_stats_S1.csv
t=$(find gpurun_out/trso -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/frame_timeline.py $t gpurun_out/so_frame_timeline_S1.txt
rm -rf gpurun_out/trso
tail -2 gpurun_out/so_trace.log; head -70 gpurun_out/so_frame_timeline_S1.txt
