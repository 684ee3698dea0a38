#!/bin/bash
# Round 5 closing session: PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) of the headline command - conv and GroupNorm families - bound to the
# kernel-source hash, kernel trace / stats / frame timelines (32 sequences, one sequence), the driver's bench command, the 4096^3
# calibration, the host-issue floor, the whole GPU test suite.  Nothing under dynaboa_amd/csrc may change after this session (bench.py
# refuses PMC summaries of other sources).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
HASH=$(python -c "import bench; print(bench.csrc_sha16())")
echo "csrc hash $HASH"
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trf -o trace -- python $R/bench.py --seqs 32 --steps 8 --warmup 2 $Q) > $O/trace.log 2>&1
f=$(find $O/trf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S32.csv && python tools/step_breakdown.py $O/kernel_stats_S32.csv 10 | tee $O/step_breakdown_S32.txt
t=$(find $O/trf -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S32.txt && head -6 $O/frame_timeline_S32.txt
rm -rf $O/trf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr1 -o trace -- python $R/bench.py --seqs 1 --steps 8 --warmup 2 $Q) > $O/trace_S1.log 2>&1
f=$(find $O/tr1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_S1.csv
t=$(find $O/tr1 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/frame_timeline.py $t $O/frame_timeline_S1.txt && head -5 $O/frame_timeline_S1.txt
rm -rf $O/tr1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > $O/pmc_$C.log 2>&1
  f=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summarize.py $f $C $O/pmc_$C.json > $O/pmc_$C.txt 2>&1
  rm -rf $O/pmc_$C
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json "round 5 closing session" $O/pmc_igemm_traffic.json 32 $HASH > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt
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
cp $O/pmc_igemm_traffic.json profiles/r05_pmc_igemm_traffic.json
cp $O/pmc_sq_tp.json profiles/r05_pmc_sq_tp.json
cp $O/pmc_gn_traffic.json profiles/r05_pmc_gn_traffic.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --conv_table $O/conv_table_S32.csv > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python tools/conv_table.py $O/conv_table_S32.csv 80 > $O/conv_table_S32.txt 2>/dev/null; head -7 $O/conv_table_S32.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "S1", d.get("single_stream_frames_per_s"), "SO", d.get("second_order_single_stream_frames_per_s"),
      "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic","conv_ms_per_step","conv_busy_ms_per_step")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"))
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
timeout 200 python tools/tp_lab.py 16 16 4096 4096 1 1 2>/dev/null | tail -1 > $O/gemm4096.json; cat $O/gemm4096.json
timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee $O/host_floor.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
DYB_BENCH_SMOKE_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no_cpu_baseline > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_two_ranks_one_gpu.json").read().strip().splitlines()[-1])
    print("two ranks on one GPU (control flow only):", d["n_gpus"], round(d["value"], 1), (d.get("pw3d_operating_point") or {}).get("value"))
except Exception as e:
    print("two-rank smoke failed:", e, open("$O/bench_two_ranks_one_gpu.err").read()[-600:])
PY
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
