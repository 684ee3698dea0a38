#!/bin/bash
# Round 4, GPU session 5: defaults after the GroupNorm lab (1024-work-item one-pass, poll spacing 8), few sequences per GPU as
# several free-running groups (own stream + issuing thread each) instead of one lockstep group, one sequence alone.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), "groups", d["config"].get("lockstep_groups"), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
one s32 "" "--seqs 32 --steps 12 --warmup 3"
one s32b "" "--seqs 32 --steps 12 --warmup 3"
one s1 "" "--seqs 1 --steps 40 --warmup 8"
one s2_g2 "" "--seqs 2 --groups 2 --steps 30 --warmup 6"
one s4_g1 "" "--seqs 4 --groups 1 --steps 20 --warmup 5"
one s4_g2 "" "--seqs 4 --groups 2 --steps 20 --warmup 5"
one s4_g4 "" "--seqs 4 --groups 4 --steps 20 --warmup 5"
one s5_g5 "" "--seqs 5 --groups 5 --steps 20 --warmup 5"
one s8_g2 "" "--seqs 8 --groups 2 --steps 16 --warmup 4"
one s8_g4 "" "--seqs 8 --groups 4 --steps 16 --warmup 4"
one s8_g8 "" "--seqs 8 --groups 8 --steps 16 --warmup 4"
one s16_g2 "" "--seqs 16 --groups 2 --steps 12 --warmup 3"
one s16_g4 "" "--seqs 16 --groups 4 --steps 12 --warmup 3"
