#!/bin/bash
# r06 final2: the driver's bench command and the GPU suite once more at HEAD (bench.py changed after the closing session: the PCIe-inclusive pass
# times 20 steps; csrc unchanged - hash 57c7ff102492840a, the committed PMC summaries stay valid)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/final2; O=gpurun_out/final2; export TMPDIR=/tmp
python -c "import bench; print('csrc hash', bench.csrc_sha16())"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "S1", d.get("single_stream_frames_per_s"), "SO", d.get("second_order_single_stream_frames_per_s"),
      "roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_while_convs_run","frac_while_convs_run","traffic","whole_frame_frac")})
for k,v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, v.get("value"), v.get("ms_per_step"), v.get("dynamic_loop_extra_steps_mean"), v.get("error"))
sw = d.get("sequences_per_gpu_sweep", {})
print("sweep", {k: (round(v["value"],1) if isinstance(v, dict) and v.get("value") else None) for k, v in sw.items() if k != "note"})
b = d.get("batch16_fp32_vs_bf16", {})
print("b16", {k: (v or {}).get("value") for k, v in b.items()}, (d.get("batch16_first_vs_second_order", {}).get("second_order") or {}).get("value"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
