#!/bin/bash
# Round 4, GPU session 10: the two second-order side runs that fell 4x in the closing session, with the chain stream at normal and at high priority.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for P in 0 -1 0 -1; do
DYB_CHAIN_PRIORITY=$P timeout 300 python - <<'PY' 2>&1 | grep -v Warning | tail -4
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
print("priority", os.environ.get("DYB_CHAIN_PRIORITY"))
r = bench.sub_record(dev, "so_fd", 12, 3, 1, 3, "fd", second_order=1, hvp="fd")
print("second order fd:", r.get("value"), r.get("ms_per_step"), r.get("host_issue_ms_per_step"), r.get("error"))
r = bench.sub_record(dev, "so_full", 4, 1, 1, 1, "so full", full_losses=1, second_order=1, hvp="exact", hvp_terms="all")
print("second order full:", r.get("value"), r.get("ms_per_step"), r.get("host_issue_ms_per_step"), r.get("error"))
r = bench.sub_record(dev, "so", 8, 2, 1, 3, "so", second_order=1, hvp="exact")
print("second order exact:", r.get("value"), r.get("ms_per_step"), r.get("host_issue_ms_per_step"), r.get("error"))
PY
done
