#!/bin/bash
# Round 4, GPU session 8: dispatch priority of the chain's stream (the auxiliary stream has slack), the two relaxed bounds again, the dynamic-loop records.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
Q="--no_cpu_baseline --no_sub_records --percentile_frames 0"
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step | conv per-launch TF", round(r.get("achieved", 0), 1), "union TF", round(r.get("achieved_while_convs_run") or 0, 1),
          "conv ms/step", round(r.get("conv_ms_per_step", 0), 1), "busy", round(r.get("conv_busy_ms_per_step") or 0, 1), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one prio0 "DYB_CHAIN_PRIORITY=0" "--seqs 32 --steps 12 --warmup 3"
one prio_hi "DYB_CHAIN_PRIORITY=-1" "--seqs 32 --steps 12 --warmup 3"
one prio0_b "DYB_CHAIN_PRIORITY=0" "--seqs 32 --steps 12 --warmup 3"
one prio_hi_b "DYB_CHAIN_PRIORITY=-1" "--seqs 32 --steps 12 --warmup 3"
one prio_hi_s1 "DYB_CHAIN_PRIORITY=-1" "--seqs 1 --steps 40 --warmup 8"
timeout 600 python -m pytest tests/test_adaptation_gpu.py tests/test_headline_gpu.py -q -m gpu -x -k "fused_level_node or headline_schedule" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python - <<'PY' 2>&1 | grep -v Warning | tail -6
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
thr, tab = bench.calibrate_gate_threshold(dev)
print("calibrated threshold", thr, "probes", tab["probes"])
r = bench.sub_record(dev, "dyn_S1", 16, 4, 1, 1, "dynamic loop entered, one sequence", full_losses=1, cos_sim_threshold=thr)
print("dynamic S1:", r.get("value"), r.get("ms_per_step"), r.get("dynamic_loop_extra_steps_mean"), r.get("error"))
r = bench.sub_record(dev, "dyn_S32", 8, 2, 1, 1, "dynamic loop entered, 32 sequences", seqs=32, full_losses=1, cos_sim_threshold=thr)
print("dynamic S32:", r.get("value"), r.get("ms_per_step"), r.get("dynamic_loop_extra_steps_mean"), r.get("error"))
PY
