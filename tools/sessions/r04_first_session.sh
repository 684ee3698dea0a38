#!/bin/bash
# Round 4, first GPU session (prepared at the end of round 3, not yet run): what round 3 left unmeasured.
#   1. sequences per GPU beyond 32 (the stepper takes up to 64 replicas): 32 / 48 / 64
#   2. the second-order frame at HEAD (pair-add fusion, batched head, batched multi-pass head: parity checked, not timed)
#   3. the headline with the batched per-frame bookkeeping / metric flush (host side, not timed)
# ~4 GPU-minutes.  Follow with tools/r03_final.sh-style PMC passes once kernel sources change.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
for S in 32 48 64; do
  timeout 200 python bench.py --seqs $S --steps 10 --warmup 3 $Q > gpurun_out/r04_seqs$S.json 2> gpurun_out/r04_seqs$S.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_seqs$S.json").read().strip().splitlines()[-1])
    print("seqs $S:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d["host_issue_ms_per_step"], 2))
except Exception as e:
    print("seqs $S failed:", e, open("gpurun_out/r04_seqs$S.err").read()[-400:])
PY
done
timeout 280 python - <<'PY' 2>&1 | grep -v Warning | tail -8
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
r = bench.sub_record(dev, "so_s1", 16, 4, 1, 3, "one sequence second order exact hvp", second_order=1, hvp="exact")
print("second order, one sequence:", round(r["value"], 2), "frames/s", round(r["ms_per_step"], 2), "ms", flush=True)
r = bench.sub_record(dev, "so_full", 8, 2, 1, 1, "default term set second order", full_losses=1, second_order=1, hvp="exact", hvp_terms="all")
print("second order, default term set:", round(r["value"], 2), "frames/s", flush=True)
r = bench.sub_record(dev, "b16_so", 6, 2, 16, 3, "b16 second order exact hvp", second_order=1, hvp="exact")
print("second order, batch 16:", round(r["value"], 2), "frames/s", flush=True)
PY
