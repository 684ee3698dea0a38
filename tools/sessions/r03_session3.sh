#!/bin/bash
# round 3, GPU session 3: two K-steps of loads in flight (tp_kernel 3) vs one (2); SQ counters of both (MFMA pipe busy, wait classes, clock)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--steps 12 --warmup 4 --no_cpu_baseline --no_sub_records --percentile_frames 0"
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py $ARGS --conv_table gpurun_out/s3_table_$tag.csv > gpurun_out/s3_bench_$tag.json 2> gpurun_out/s3_bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s3_bench_$tag.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("$tag", "frames/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "conv TF", round(r.get("achieved",0),1), "conv ms/step", round(r.get("conv_ms_per_step",0),2))
except Exception as e: print("$tag FAILED", e)
PY
  python tools/conv_table.py gpurun_out/s3_table_$tag.csv 70 > gpurun_out/s3_table_$tag.txt 2>/dev/null
  head -6 gpurun_out/s3_table_$tag.txt
}
run pipe2 DYB_TP_KERNEL=3
run pipe2_g768 DYB_TP_KERNEL=3 DYB_TP_GRID=768
run pipe1_g640 DYB_TP_KERNEL=2 DYB_TP_GRID=640
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
for K in 2 3; do
  (cd /tmp && DYB_TP_KERNEL=$K timeout 400 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_sq$K -o pmc -- python $R/bench.py --seqs 32 --steps 2 --warmup 1 $Q) > gpurun_out/s3_pmc_sq$K.log 2>&1
  f=$(find gpurun_out/pmc_sq$K -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_multi.py $f gpurun_out/s3_pmc_sq_tpk$K.json igemm > gpurun_out/s3_pmc_sq_tpk$K.txt 2>&1
  rm -rf gpurun_out/pmc_sq$K
  head -50 gpurun_out/s3_pmc_sq_tpk$K.txt
done
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "throughput and pipelined2" > gpurun_out/s3_pytest_kernels.txt 2>&1; tail -3 gpurun_out/s3_pytest_kernels.txt
