#!/bin/bash
# r05 s14: where does a step go at 5 sequences per GPU (the 3DPW operating point of an 8-GPU node)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s14; O=gpurun_out/s14; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0"
for S in 5 8; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr$S -o trace -- python $R/bench.py --seqs $S --steps 8 --warmup 2 $Q --no_roofline) > $O/trace_S$S.log 2>&1
f=$(find $O/tr$S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S$S.csv
t=$(find $O/tr$S -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S$S.txt && head -4 $O/frame_timeline_S$S.txt
rm -rf $O/tr$S
timeout 200 python bench.py --seqs $S --steps 16 --warmup 4 $Q --conv_table $O/ct_S$S.csv > $O/b_S$S.json 2> $O/b_S$S.err
python tools/conv_table.py $O/ct_S$S.csv 80 > $O/ct_S$S.txt 2>/dev/null; head -7 $O/ct_S$S.txt
done
for tm in 4 5; do DYB_TP_MIN=$tm timeout 200 python bench.py --seqs 5 --steps 16 --warmup 4 $Q --no_roofline > $O/b_S5_tpmin$tm.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("$O/b_S5_tpmin$tm.json").read().strip().splitlines()[-1]); print("S5 tp_min=$tm", round(d["value"], 1))
PY
done
