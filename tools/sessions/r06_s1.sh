#!/bin/bash
# r06 s1: first GPU look at round 6 - the whole parity suite (new: gate goldens at the literal defaults, noise-derived end-of-stream bounds,
# share_dyn_fwd identity, two-rank control flow on one GPU), smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s1; O=gpurun_out/s1; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -x -s 2>&1 | grep -v "^$" | tail -60 > $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -25 $O/pytest_gpu.log; tail -1 $O/smoke.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/s1/bench_default.json"))
print("headline", round(d["value"],1), d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved","frac","traffic")})
print("single", d.get("single_stream_frames_per_s"), json.dumps(d["single_stream"].get("roofline"))[:400], json.dumps(d["single_stream"].get("hbm_view"))[:200])
for k in ("second_order","batch8_exemplars","full_default_losses","full_default_losses_S32","full_default_losses_S32_exemplars_resident","full_default_losses_dynamic","full_default_losses_dynamic_S32","second_order_full_losses_exact_hvp"):
    v=d.get(k) or {}
    print(k, v.get("value"), v.get("error"))
print("b16", {k:(v.get("value") if isinstance(v,dict) else v) for k,v in d["batch16_fp32_vs_bf16"].items()})
print("cpu", json.dumps(d["cpu_baseline"])[:300])
PY
