#!/bin/bash
# second-order session 3: overlap settings not yet measured (one sequence: 2 again, 6, 3; batch 16: 2, 6)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 280 python - <<'PY' 2>&1 | grep -v Warning | tail -12
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
for ov in (2, 6, 3, 0, 2, 7):
    os.environ["DYB_HVP_OVERLAP"] = str(ov)
    r = bench.sub_record(dev, "so_s1", 16, 4, 1, 3, "one sequence second order exact hvp", second_order=1, hvp="exact")
    print("S1 overlap", ov, round(r["value"], 2), "frames/s", round(r["ms_per_step"], 2), "ms host", round(r["host_issue_ms_per_step"], 2), flush=True)
for ov in (2, 6, 7):
    os.environ["DYB_HVP_OVERLAP"] = str(ov)
    r = bench.sub_record(dev, "b16_so", 6, 2, 16, 3, "b16 second order exact hvp", second_order=1, hvp="exact")
    print("B16 overlap", ov, round(r["value"], 2), "frames/s", round(r["ms_per_step"], 2), "ms host", round(r["host_issue_ms_per_step"], 2), flush=True)
PY
