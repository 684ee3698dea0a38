#!/bin/bash
# r05 s12 (diagnostic, results of the second run are wrong by design): what does the producer-statistics fold in every consumer
# workgroup's prologue cost?  one-stream conv table with the fused-loader forwards reading ONE record instead of all
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s12; O=gpurun_out/s12; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0"
for tag in full one; do
  E="X=1"; [ $tag = one ] && E="DYB_DIAG_FA_ONE_RECORD=1"
  env DYB_NO_AUX=1 DYB_UPD_OVERLAP=0 $E timeout 300 python bench.py --seqs 32 --steps 8 --warmup 3 $Q --conv_table $O/ct_$tag.csv > $O/b_$tag.json 2> $O/b_$tag.err
  python tools/conv_table.py $O/ct_$tag.csv 80 > $O/ct_$tag.txt 2>/dev/null; head -6 $O/ct_$tag.txt
  env $E timeout 300 python bench.py --seqs 32 --steps 10 --warmup 3 $Q --no_roofline > $O/h_$tag.json 2> $O/h_$tag.err
  python - <<PY
import json
d = json.loads(open("$O/h_$tag.json").read().strip().splitlines()[-1]); print("$tag two-queue headline", round(d["value"], 1))
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); print("$tag one-stream", round(d["value"], 1), round(d["ms_per_step"], 2))
PY
done
env DYB_DIAG_FA_ONE_RECORD=1 timeout 200 python bench.py --seqs 1 --steps 40 --warmup 8 $Q --no_roofline > $O/s1_one.json 2>/dev/null
timeout 200 python bench.py --seqs 1 --steps 40 --warmup 8 $Q --no_roofline > $O/s1_full.json 2>/dev/null
python - <<PY
import json
for t in ("full", "one"):
    d = json.loads(open("$O/s1_%s.json" % t).read().strip().splitlines()[-1]); print("one sequence,", t, round(d["value"], 1))
PY
