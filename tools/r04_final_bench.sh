#!/bin/bash
# Round 4: the driver's bench command once more at the final sources (the closing session's record was taken with the chain stream at
# high priority, which broke two second-order side runs: see bench.chain_stream); PMC summaries under profiles/ stay valid (same csrc hash).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/final7; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --conv_table $O/conv_table_S32.csv > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python tools/conv_table.py $O/conv_table_S32.csv 80 > $O/conv_table_S32.txt 2>/dev/null; head -7 $O/conv_table_S32.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic","conv_ms_per_step","conv_busy_ms_per_step")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else None) for k, v in sw.items() if k != "note"})
print("b16", {k: v.get("value") for k, v in d["batch16_fp32_vs_bf16"].items()}, d["batch16_first_vs_second_order"]["second_order"].get("value"))
PY
