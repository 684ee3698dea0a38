#!/bin/bash
# Round 6 closing session: PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) of the headline command - conv and GroupNorm families - bound to the
# kernel-source hash, kernel trace / stats / frame timelines (32 sequences, one sequence), conv tables (in the path / one stream), the
# driver's bench command, the 4096^3 calibration, the host-issue floor, smoke, the two-rank control flow, the whole GPU test suite.
# Nothing under dynaboa_amd/csrc may change after this session (bench.py refuses PMC summaries of other sources).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
HASH=$(python -c "import bench; print(bench.csrc_sha16())")
echo "csrc hash $HASH"
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32.csv && python tools/step_breakdown.py $O/kernel_stats_S32.csv 10 | tee $O/step_breakdown_S32.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt > /dev/null 2>&1 && head -6 $O/frame_timeline_S32.txt
rm -rf $O/trf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr1 -o trace -- python $R/bench.py --seqs 1 --steps 8 --warmup 2 $Q) > $O/trace_S1.log 2>&1
f=$(find $O/tr1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S1.csv
t=$(find $O/tr1 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S1.txt > /dev/null 2>&1 && head -5 $O/frame_timeline_S1.txt
rm -rf $O/tr1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > $O/pmc_$C.log 2>&1
  f=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C $O/pmc_$C.json > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json "round 6 closing session" $O/pmc_igemm_traffic.json 32 $HASH > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
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
cp $O/pmc_igemm_traffic.json profiles/r06_pmc_igemm_traffic.json
cp $O/pmc_sq_tp.json profiles/r06_pmc_sq_tp.json
cp $O/pmc_gn_traffic.json profiles/r06_pmc_gn_traffic.json
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --conv_table $O/conv_table_S32.csv > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python tools/conv_table.py $O/conv_table_S32.csv 80 > $O/conv_table_S32.txt 2>/dev/null; head -7 $O/conv_table_S32.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "S1", d.get("single_stream_frames_per_s"), "SO", d.get("second_order_single_stream_frames_per_s"),
      "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic","conv_ms_per_step","conv_busy_ms_per_step","whole_frame_frac")})
print("single_stream roofline", json.dumps((d.get("single_stream") or {}).get("roofline"))[:600])
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"), v.get("error"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else None) for k, v in sw.items() if k != "note"})
b = d.get("batch16_fp32_vs_bf16", {})
print("b16", {k: (v or {}).get("value") for k, v in b.items()}, (d.get("batch16_first_vs_second_order", {}).get("second_order") or {}).get("value"))
PY
# the same headline loop with every launch on ONE stream: what the kernels cost without company (conv family's rate alone)
DYB_NO_AUX=1 DYB_UPD_OVERLAP=0 timeout 300 python bench.py --seqs 32 --steps 8 --warmup 3 --no_cpu_baseline --no_sub_records --percentile_frames 0 --conv_table $O/conv_table_one_stream.csv > $O/bench_one_stream.json 2> $O/bench_one_stream.err
python tools/conv_table.py $O/conv_table_one_stream.csv 80 > $O/conv_table_one_stream.txt 2>/dev/null; head -7 $O/conv_table_one_stream.txt
python - <<PY
import json
d=json.loads(open("$O/bench_one_stream.json").read().strip().splitlines()[-1])
print("one stream:", round(d["value"],1), "frames/s", round(d["ms_per_step"],2), "ms/step, conv", d["roofline"].get("achieved"), d["roofline"].get("frac"))
PY
# round 5's form of the weight updates (every gradient to HBM, streaming fast-weight / Adam passes over the whole arena): the same-session A/B
DYB_FUSE_FAST=0 DYB_FUSE_ADAM=0 DYB_FUSE_LINEAR=0 DYB_FUSE_EMA=0 DYB_TP_STEM=0 timeout 300 python bench.py --seqs 32 --steps 20 --warmup 5 --no_cpu_baseline --no_sub_records --percentile_frames 0 > $O/bench_round5_updates.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/bench_round5_updates.json").read().strip().splitlines()[-1])
print("round-5 update form:", round(d["value"],1), "frames/s", round(d["ms_per_step"],2), "ms/step, conv", d["roofline"].get("achieved"), d["roofline"].get("frac"))
PY
timeout 200 python tools/tp_lab.py 16 16 4096 4096 1 1 2>/dev/null | tail -1 > $O/gemm4096.json; cat $O/gemm4096.json
timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee $O/host_floor.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
