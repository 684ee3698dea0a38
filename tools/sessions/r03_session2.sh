#!/bin/bash
# round 3, GPU session 2: A/B of the two loop forms of igemm_tp_kernel at 32 sequences (bench + per-shape conv table), split sweep,
# kernel parity tests of the new form
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--steps 12 --warmup 4 --no_cpu_baseline --no_sub_records --percentile_frames 0"
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py $ARGS --conv_table gpurun_out/s2_table_$tag.csv > gpurun_out/s2_bench_$tag.json 2> gpurun_out/s2_bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s2_bench_$tag.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("$tag", "frames/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "conv TF", round(r.get("achieved",0),1), "conv ms/step", round(r.get("conv_ms_per_step",0),2))
except Exception as e: print("$tag FAILED", e)
PY
  python tools/conv_table.py gpurun_out/s2_table_$tag.csv 70 > gpurun_out/s2_table_$tag.txt 2>/dev/null
  head -6 gpurun_out/s2_table_$tag.txt
}
run phased DYB_TP_KERNEL=1
run pipe DYB_TP_KERNEL=2
run pipe_g384 DYB_TP_KERNEL=2 DYB_TP_GRID=384
run pipe_g768 DYB_TP_KERNEL=2 DYB_TP_GRID=768
run pipe_g1024 DYB_TP_KERNEL=2 DYB_TP_GRID=1024
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "throughput" > gpurun_out/s2_pytest_kernels.txt 2>&1; tail -3 gpurun_out/s2_pytest_kernels.txt
timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_adaptation_gpu.py -q -m gpu > gpurun_out/s2_pytest_e2e.txt 2>&1; tail -5 gpurun_out/s2_pytest_e2e.txt
