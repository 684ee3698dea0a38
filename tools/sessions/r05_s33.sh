#!/bin/bash
# r05 s33: queue timelines of the default term set with the level's passes side by side (one sequence, 32 sequences; 8 hardware queues)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s33; O=gpurun_out/s33; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
F="--full_losses 1 --inner_step 1 --seqs_full 1"
trace() { # tag seqs steps warm
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$1 -o trace -- python $R/bench.py --seqs $2 --steps $3 --warmup $4 $Q $F) > $O/trace_$1.log 2>&1
  t=$(find $O/tr_$1 -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_full_$1.txt && head -7 $O/frame_timeline_full_$1.txt
  rm -rf $O/tr_$1
}
trace S32 32 4 1
trace S1 1 10 3
