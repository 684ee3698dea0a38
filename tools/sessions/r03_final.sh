#!/bin/bash
# round 3 closing session: GPU test suite, the driver's bench command, kernel trace / stats, PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) - all at
# the same kernel sources (their hash goes into the PMC summaries; bench.py refuses summaries of other sources)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
HASH=$(python -c "import bench; print(bench.csrc_sha16())")
echo "csrc hash $HASH"
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > gpurun_out/f_pmc_$C.log 2>&1
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C gpurun_out/f_pmc_$C.json > gpurun_out/f_pmc_$C.txt 2>&1
  rm -rf gpurun_out/pmc_$C
done
python tools/pmc_traffic.py gpurun_out/f_pmc_FETCH_SIZE.json gpurun_out/f_pmc_WRITE_SIZE.json "round 3 closing session" gpurun_out/f_pmc_igemm_traffic.json 32 $HASH > gpurun_out/f_pmc_traffic.txt 2>&1
cat gpurun_out/f_pmc_traffic.txt
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
(cd /tmp && timeout 400 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_sq -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > gpurun_out/f_pmc_sq.log 2>&1
f=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_multi.py $f gpurun_out/f_pmc_sq_tp.json igemm > gpurun_out/f_pmc_sq_tp.txt 2>&1
python - <<PY
import json
p="gpurun_out/f_pmc_sq_tp.json"
d=json.load(open(p)); d["_csrc_sha16"]="$HASH"; json.dump(d, open(p,"w"), indent=1)
PY
rm -rf gpurun_out/pmc_sq
head -12 gpurun_out/f_pmc_sq_tp.txt
# the PMC summaries become visible to bench.py (profiles/ of this checkout)
cp gpurun_out/f_pmc_igemm_traffic.json profiles/r03_pmc_igemm_traffic.json
cp gpurun_out/f_pmc_sq_tp.json profiles/r03_pmc_sq_tp.json
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trf -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > gpurun_out/f_trace.log 2>&1
f=$(find gpurun_out/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/f_kernel_stats_S32.csv
t=$(find gpurun_out/trf -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/frame_timeline.py $t gpurun_out/f_frame_timeline_S32.txt
rm -rf gpurun_out/trf
head -8 gpurun_out/f_frame_timeline_S32.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 --conv_table gpurun_out/f_conv_table_S32.csv > gpurun_out/f_bench_default.json 2> gpurun_out/f_bench_default.err ) 2> gpurun_out/f_bench_time.txt
tail -3 gpurun_out/f_bench_time.txt
python tools/conv_table.py gpurun_out/f_conv_table_S32.csv 80 > gpurun_out/f_conv_table_S32.txt 2>/dev/null; head -6 gpurun_out/f_conv_table_S32.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/f_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","traffic","conv_ms_per_step")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"))
PY
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/f_pytest_gpu.txt 2>&1; tail -4 gpurun_out/f_pytest_gpu.txt
