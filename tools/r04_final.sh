#!/bin/bash
# Round 4 closing session: PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) of the headline command - conv family AND GroupNorm family - bound to
# the kernel-source hash, kernel trace / stats / frame timeline, the driver's bench command, the whole GPU test suite.  Nothing under
# dynaboa_amd/csrc may change after this session (bench.py refuses PMC summaries of other sources).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
HASH=$(python -c "import bench; print(bench.csrc_sha16())")
echo "csrc hash $HASH"
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32.csv && python tools/step_breakdown.py $O/kernel_stats_S32.csv 10 | tee $O/step_breakdown_S32.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt && head -6 $O/frame_timeline_S32.txt
[ -n "$t" ] && python tools/trace_by_grid.py $t "gn_" > $O/gn_by_grid.txt
rm -rf $O/trf
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > $O/pmc_$C.log 2>&1
  f=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C $O/pmc_$C.json > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json "round 4 closing session" $O/pmc_igemm_traffic.json 32 $HASH > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
python tools/pmc_family.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/kernel_stats_S32.csv gn_ 32 $O/pmc_gn_traffic.json 2>&1 | tee $O/pmc_gn_traffic.txt | tail -8
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
(cd /tmp && timeout 400 rocprofv3 --pmc $CNT --output-format csv -d $R/$O/pmc_sq -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > $O/pmc_sq.log 2>&1
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_multi.py $f $O/pmc_sq_tp.json igemm > $O/pmc_sq_tp.txt 2>&1
python - <<PY
import json
p="$O/pmc_sq_tp.json"
try:
    d=json.load(open(p)); d["_csrc_sha16"]="$HASH"; json.dump(d, open(p,"w"), indent=1)
except Exception as e:
    print("sq summary missing:", e)
PY
rm -rf $O/pmc_sq
head -8 $O/pmc_sq_tp.txt
# the PMC summaries become visible to bench.py (profiles/ of this checkout)
cp $O/pmc_igemm_traffic.json profiles/r04_pmc_igemm_traffic.json
cp $O/pmc_sq_tp.json profiles/r04_pmc_sq_tp.json
cp $O/pmc_gn_traffic.json profiles/r04_pmc_gn_traffic.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --conv_table $O/conv_table_S32.csv > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python tools/conv_table.py $O/conv_table_S32.csv 80 > $O/conv_table_S32.txt 2>/dev/null; head -7 $O/conv_table_S32.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic","conv_ms_per_step","conv_busy_ms_per_step")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else None) for k, v in sw.items() if k != "note"})
PY
# one sequence: the side stream's launches from the library's helper thread (default) against in line
for tag in on off on2 off2; do
  case $tag in on*) E=DYB_SIDE_THREAD=1;; *) E=DYB_SIDE_THREAD=0;; esac
  env $E timeout 200 python bench.py --seqs 1 --steps 60 --warmup 10 $Q > $O/bench_s1_$tag.json 2> $O/bench_s1_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_s1_$tag.json").read().strip().splitlines()[-1])
    print("one sequence, helper thread $tag:", round(d["value"], 2), "frames/s", round(d["ms_per_step"], 3), "ms/frame, host issue", round(d.get("host_issue_ms_per_step", 0), 3), flush=True)
except Exception as e:
    print("s1 $tag failed:", e, open("$O/bench_s1_$tag.err").read()[-500:])
PY
done
# second order, one sequence: launches per frame with the tangent pairs as one launch (kernel stats of 8 frames)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trso -o trace -- python $R/bench.py --second_order 1 --seqs 1 --steps 8 --warmup 2 $Q) > $O/trace_so.log 2>&1
f=$(find $O/trso -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_SO_S1.csv && python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_SO_S1.csv")))
calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("second order, one sequence: %d launches, %.1f ms of kernels over the run (10 frames incl. warm-up: %.0f launches per frame)" % (calls, tot / 1e6, calls / 10.0))
PY
rm -rf $O/trso
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
