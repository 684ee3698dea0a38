#!/bin/bash
# r05 s26: the driver's bench command with 4 hardware queues (the closing session's run with 8 showed late side runs collapsing)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s26; O=gpurun_out/s26; export TMPDIR=/tmp
( time GPU_MAX_HW_QUEUES=4 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "S1", d.get("single_stream_frames_per_s"), "SO", d.get("second_order_single_stream_frames_per_s"),
      "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else None) for k, v in sw.items() if k != "note"})
b = d.get("batch16_fp32_vs_bf16", {})
print("b16", {k: (v or {}).get("value") for k, v in b.items()}, (d.get("batch16_first_vs_second_order", {}).get("second_order") or {}).get("value"))
PY
