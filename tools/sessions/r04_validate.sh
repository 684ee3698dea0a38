#!/bin/bash
# Round 4: mid-round validation - the whole GPU test suite and the driver's bench command at HEAD (no counters).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/v1; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --conv_table $O/conv_table_S32.csv > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python tools/conv_table.py $O/conv_table_S32.csv 80 > $O/conv_table_S32.txt 2>/dev/null; head -7 $O/conv_table_S32.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2))
print("roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","conv_ms_per_step","conv_busy_ms_per_step")})
print("pcie", d.get("pcie_inclusive", {}).get("value"))
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"), v.get("error"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else v.get("error") if isinstance(v, dict) else None) for k, v in sw.items() if k != "note"})
PY
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
