#!/bin/bash
# Round 4, GPU session 7: re-sweep of the split / chunking targets now that the GroupNorm work left the chain (tp_grid, tp_gn_wgs), the new GPU
# test, the dynamic-loop calibration, S = 37 (every 3DPW test track on one GPU).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
timeout 600 python -m pytest tests/test_headline_gpu.py -q -m gpu -x -k "ranged" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step", flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one base "" "--seqs 32 --steps 12 --warmup 3"
one grid384 "DYB_TP_GRID=384" "--seqs 32 --steps 12 --warmup 3"
one grid640 "DYB_TP_GRID=640" "--seqs 32 --steps 12 --warmup 3"
one grid768 "DYB_TP_GRID=768" "--seqs 32 --steps 12 --warmup 3"
one gnwgs512 "DYB_TP_GN_WGS=512" "--seqs 32 --steps 12 --warmup 3"
one gnwgs2048 "DYB_TP_GN_WGS=2048" "--seqs 32 --steps 12 --warmup 3"
one base2 "" "--seqs 32 --steps 12 --warmup 3"
one s37 "" "--seqs 37 --steps 12 --warmup 3"
timeout 300 python - <<'PY' 2>&1 | grep -v Warning | tail -6
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
thr, tab = bench.calibrate_gate_threshold(dev)
print("calibrated threshold", thr, "probes", tab["probes"])
r = bench.sub_record(dev, "dyn_S32", 8, 2, 1, 1, "dynamic loop entered, 32 sequences", seqs=32, full_losses=1, cos_sim_threshold=thr)
print("dynamic S32:", r.get("value"), r.get("ms_per_step"), r.get("dynamic_loop_extra_steps_mean"), r.get("error"))
r = bench.sub_record(dev, "dyn_S1", 16, 4, 1, 1, "dynamic loop entered, one sequence", full_losses=1, cos_sim_threshold=thr)
print("dynamic S1:", r.get("value"), r.get("ms_per_step"), r.get("dynamic_loop_extra_steps_mean"), r.get("error"))
PY
