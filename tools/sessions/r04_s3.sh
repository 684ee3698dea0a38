#!/bin/bash
# Round 4, GPU session 3: one-pass GroupNorm backward by workgroup size (256 / 512 / 1024), ranged weight updates beside the forward
# (upd_overlap on / off), the throughput schedule's threshold at few sequences per GPU (tp_min at S = 4, 5, 8), kernel durations by grid.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_headline_gpu.py -q -m gpu -s -k "onepass or one_image_throughput or headline" > $O/pytest.txt 2>&1; grep -E "passed|failed|first-frame outer" $O/pytest.txt | tail -6
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one t256_ovl1 "DYB_TP_GN_THREADS=256 DYB_UPD_OVERLAP=1" "--seqs 32 --steps 12 --warmup 3"
one t256_ovl0 "DYB_TP_GN_THREADS=256 DYB_UPD_OVERLAP=0" "--seqs 32 --steps 12 --warmup 3"
one t512_ovl1 "DYB_TP_GN_THREADS=512 DYB_UPD_OVERLAP=1" "--seqs 32 --steps 12 --warmup 3"
one t1024_ovl1 "DYB_TP_GN_THREADS=1024 DYB_UPD_OVERLAP=1" "--seqs 32 --steps 12 --warmup 3"
one t256_ovl1_b "DYB_TP_GN_THREADS=256 DYB_UPD_OVERLAP=1" "--seqs 32 --steps 12 --warmup 3"
one s64 "DYB_TP_GN_THREADS=256 DYB_UPD_OVERLAP=1" "--seqs 64 --steps 8 --warmup 2"
for S in 4 5 8; do
  one s${S}_tpmin8 "DYB_TP_MIN=8 DYB_TP_GN_WGS=1024" "--seqs $S --steps 16 --warmup 4"
  one s${S}_tpmin4 "DYB_TP_MIN=4" "--seqs $S --steps 16 --warmup 4"
done
one s16 "" "--seqs 16 --steps 12 --warmup 3"
echo "--- kernel trace (256-thread one-pass, ranged updates)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 6 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32.csv && python tools/step_breakdown.py $O/kernel_stats_S32.csv 8 | tee $O/step_breakdown.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_by_grid.py $t gn_bwd | tee $O/gn_bwd_by_grid.txt | head -16
[ -n "$t" ] && python tools/trace_by_grid.py $t "gn_" > $O/gn_all_by_grid.txt
[ -n "$t" ] && python tools/trace_by_grid.py $t "fastweight" | tail -4
[ -n "$t" ] && python tools/trace_by_grid.py $t "adam" | tail -4
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt && head -8 $O/frame_timeline_S32.txt
rm -rf $O/trf
