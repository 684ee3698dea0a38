#!/bin/bash
# second-order session 2: parity tests touched since session 1, the one-sequence second-order frame and the batch-16 arm per overlap setting
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -s -k "groupnorm_tangent or hessian or exact_hvp or second_order_full_loss_set or native_full_term_set or second_order_inner3" > gpurun_out/so2_pytest.txt 2>&1; tail -3 gpurun_out/so2_pytest.txt
grep -h "SO full set\|exact SO\|native vs autograd\|tstate" gpurun_out/so2_pytest.txt | cut -c1-230
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for OV in 0 7 2; do
  DYB_HVP_OVERLAP=$OV timeout 120 python bench.py --seqs 1 --second_order 1 --hvp exact --steps 16 --warmup 4 $Q > gpurun_out/so2_bench_ov$OV.json 2> gpurun_out/so2_bench_ov$OV.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/so2_bench_ov$OV.json").read().strip().splitlines()[-1])
    print("overlap $OV:", round(d["value"], 2), "frames/s", round(d["ms_per_step"], 2), "ms, host issue", round(d["host_issue_ms_per_step"], 2))
except Exception as e:
    print("overlap $OV failed", e)
PY
done
for OV in 0 7; do
DYB_HVP_OVERLAP=$OV timeout 200 python - <<'PY' > gpurun_out/so2_b16_ov$OV.txt 2>&1
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
r = bench.sub_record(dev, "b16_so", 6, 2, 16, 3, "b16 second order exact hvp", second_order=1, hvp="exact")
print("b16 second order overlap", os.environ["DYB_HVP_OVERLAP"], json.dumps({k: v for k, v in r.items() if k != "config"}))
r = bench.sub_record(dev, "so_full_exact", 6, 2, 1, 1, "default term set, second order, exact hvp for every level", full_losses=1, second_order=1, hvp="exact", hvp_terms="all")
print("full-set second order overlap", os.environ["DYB_HVP_OVERLAP"], json.dumps({k: v for k, v in r.items() if k != "config"}))
PY
tail -2 gpurun_out/so2_b16_ov$OV.txt | cut -c1-300
done
