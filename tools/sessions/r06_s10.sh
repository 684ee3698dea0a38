#!/bin/bash
# r06 s10: where a step goes now (kernel trace of the headline loop at 32 sequences: family breakdown + queue timeline)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s10; O=gpurun_out/s10; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32.csv && python tools/step_breakdown.py $O/kernel_stats_S32.csv 10 | tee $O/step_breakdown_S32.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt && head -12 $O/frame_timeline_S32.txt
rm -rf $O/trf
head -40 $O/kernel_stats_S32.csv | cut -c1-160
