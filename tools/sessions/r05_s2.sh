#!/bin/bash
# r05 s2: what do the path's kernels cost ALONE?  the headline loop with every launch on one stream (no auxiliary queue, whole-arena
# updates) against the default two-queue schedule: per-shape conv table of both + kernel stats of the one-stream run
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s2; O=gpurun_out/s2; export TMPDIR=/tmp
export DYB_TP_WT=1
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0"
DYB_NO_AUX=1 DYB_UPD_OVERLAP=0 timeout 300 python bench.py --seqs 32 --steps 8 --warmup 3 $Q --conv_table $O/conv_table_alone.csv > $O/bench_alone.json 2> $O/bench_alone.err
timeout 300 python bench.py --seqs 32 --steps 8 --warmup 3 $Q --conv_table $O/conv_table_path.csv > $O/bench_path.json 2> $O/bench_path.err
for t in alone path; do python tools/conv_table.py $O/conv_table_$t.csv 80 > $O/conv_table_$t.txt 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/bench_$t.json").read().strip().splitlines()[-1])
print("$t", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step", {k: d["roofline"].get(k) for k in ("achieved", "achieved_while_convs_run", "conv_ms_per_step", "conv_busy_ms_per_step")})
PY
head -7 $O/conv_table_$t.txt; done
(cd /tmp && DYB_NO_AUX=1 DYB_UPD_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr -o trace -- python $R/bench.py --seqs 32 --steps 4 --warmup 2 $Q --no_roofline) > $O/trace.log 2>&1
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_alone.csv && python tools/step_breakdown.py $O/kernel_stats_alone.csv 6 | tee $O/step_breakdown_alone.txt
t=$(find $O/tr -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_alone.txt && head -4 $O/frame_timeline_alone.txt
rm -rf $O/tr
