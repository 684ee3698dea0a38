#!/bin/bash
# second-order session: parity tests of the tangent kernels / passes (with and without the side stream), the one-sequence second-order
# frame under the three overlap settings, its kernel trace, and the batch-16 second-order arm
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q -x -k "groupnorm_tangent or hessian or exact_hvp or second_order or native_full_term_set" > gpurun_out/so_pytest.txt 2>&1; tail -3 gpurun_out/so_pytest.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for OV in 7 0 3; do
  DYB_HVP_OVERLAP=$OV timeout 120 python bench.py --seqs 1 --second_order 1 --hvp exact --steps 12 --warmup 3 $Q > gpurun_out/so_bench_ov$OV.json 2> gpurun_out/so_bench_ov$OV.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/so_bench_ov$OV.json").read().strip().splitlines()[-1])
    print("overlap $OV:", round(d["value"], 2), "frames/s", round(d["ms_per_step"], 2), "ms, host issue", round(d["host_issue_ms_per_step"], 2))
except Exception as e:
    print("overlap $OV failed", e)
PY
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trso -o trace -- python $R/bench.py --seqs 1 --second_order 1 --hvp exact --steps 6 --warmup 2 $Q) > gpurun_out/so_trace.log 2>&1
f=$(find gpurun_out/trso -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/so2_kernel_stats_S1.csv
rm -rf gpurun_out/trso
head -12 gpurun_out/so2_kernel_stats_S1.csv | cut -c1-150
timeout 200 python - <<'PY' > gpurun_out/so_b16.txt 2>&1
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
r = bench.sub_record(dev, "b16_so", 6, 2, 16, 3, "b16 second order exact hvp", second_order=1, hvp="exact")
print("b16 second order", json.dumps({k: v for k, v in r.items() if k != "config"}))
PY
tail -2 gpurun_out/so_b16.txt
