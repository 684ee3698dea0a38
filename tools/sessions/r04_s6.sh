#!/bin/bash
# Round 4, GPU session 6: A/B of the latest switches at 32 sequences with the conv family's roofline beside the step time (workgroup cap of
# the ranged weight updates, unsplit forward with fused statistics, fc1 hoist is in), the failed headline test again, bench duration.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "headline or hmr_engine or linear" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
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
one blocks1024 "DYB_UPD_BLOCKS=1024" "--seqs 32 --steps 12 --warmup 3"
one blocks65536 "DYB_UPD_BLOCKS=65536" "--seqs 32 --steps 12 --warmup 3"
one blocks512 "DYB_UPD_BLOCKS=512" "--seqs 32 --steps 12 --warmup 3"
one blocks2048 "DYB_UPD_BLOCKS=2048" "--seqs 32 --steps 12 --warmup 3"
one ovl0 "DYB_UPD_OVERLAP=0" "--seqs 32 --steps 12 --warmup 3"
one nosplit2_0 "DYB_TP_FWD_NOSPLIT2=0" "--seqs 32 --steps 12 --warmup 3"
one onepass0 "DYB_TP_GN_ONEPASS=0 DYB_UPD_OVERLAP=0 DYB_TP_GN_FUSE_STATS=0" "--seqs 32 --steps 12 --warmup 3"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt; grep "side run" $O/bench_default.err | tr '\n' ';'
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("\nvalue", round(d["value"],1))
for k in ("full_default_losses_dynamic", "full_default_losses_dynamic_S32", "full_default_losses", "full_default_losses_S32"):
    v = d.get(k, {}); print(k, v.get("value"), v.get("dynamic_loop_extra_steps_mean"), v.get("error"))
PY
