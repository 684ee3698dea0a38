#!/bin/bash
# Round 5, last step: the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) of the headline command on the FINAL kernel sources, the whole GPU test
# suite, smoke, and the driver's bench command (tools/r05_final.sh is the full closing session; this is its counter / test / bench part)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/final2; mkdir -p $O
HASH=$(python -c "import bench; print(bench.csrc_sha16())")
echo "csrc hash $HASH"
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 6 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32.csv
rm -rf $O/trf
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > $O/pmc_$C.log 2>&1
  f=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C $O/pmc_$C.json > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json "round 5, final kernel sources" $O/pmc_igemm_traffic.json 32 $HASH > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
python tools/pmc_family.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/kernel_stats_S32.csv gn_ 32 $O/pmc_gn_traffic.json 2>&1 | tee $O/pmc_gn_traffic.txt | tail -2
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
cp $O/pmc_igemm_traffic.json profiles/r05_pmc_igemm_traffic.json
cp $O/pmc_sq_tp.json profiles/r05_pmc_sq_tp.json
cp $O/pmc_gn_traffic.json profiles/r05_pmc_gn_traffic.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "S1", d.get("single_stream_frames_per_s"), "SO", d.get("second_order_single_stream_frames_per_s"),
      "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"), v.get("error"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else None) for k, v in sw.items() if k != "note"})
b = d.get("batch16_fp32_vs_bf16", {})
print("b16", {k: (v or {}).get("value") for k, v in b.items()}, (d.get("batch16_first_vs_second_order", {}).get("second_order") or {}).get("value"))
PY
