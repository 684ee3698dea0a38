#!/bin/bash
# round 3: kernel trace of the headline configuration at HEAD: per-kernel stats + one-frame timeline (queues, gaps)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr8 -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > gpurun_out/s8_trace.log 2>&1
ls gpurun_out/tr8 | head
f=$(find gpurun_out/tr8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/s8_kernel_stats_S32.csv
t=$(find gpurun_out/tr8 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/frame_timeline.py $t gpurun_out/s8_frame_timeline_S32.txt
rm -rf gpurun_out/tr8
head -40 gpurun_out/s8_kernel_stats_S32.csv | cut -c1-150
head -30 gpurun_out/s8_frame_timeline_S32.txt
