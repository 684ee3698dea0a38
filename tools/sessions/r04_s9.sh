#!/bin/bash
# Round 4, GPU session 9: the last range of a ranged weight update issued when the consuming forward reaches layer3 (upd_late) vs at once.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
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
one late1 "DYB_UPD_LATE=1" "--seqs 32 --steps 12 --warmup 3"
one late0 "DYB_UPD_LATE=0" "--seqs 32 --steps 12 --warmup 3"
one late1_b "DYB_UPD_LATE=1" "--seqs 32 --steps 12 --warmup 3"
one late0_b "DYB_UPD_LATE=0" "--seqs 32 --steps 12 --warmup 3"
timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_replica_full_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
