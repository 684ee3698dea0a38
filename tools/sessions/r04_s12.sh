#!/bin/bash
# Round 4, GPU session 12: the tangent passes' operand pairs (conv(tx, w) + conv(x, tw) etc. as one launch) and the one-launch GroupNorm
# tangents: parity on the GPU, second-order frame rate with each switch on / off, and the N > 1 control flow of bench.py on one GPU.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "operand_pair or groupnorm_tangent or hessian" > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
one() {   # tag, env, bench args
  env $2 timeout 300 python bench.py $3 $Q > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    print("$1:", round(d["value"], 2), "frames/s", round(d["ms_per_step"], 2), "ms/step, host issue", round(d.get("host_issue_ms_per_step", 0), 2), flush=True)
except Exception as e:
    print("$1 failed:", e, open("$O/bench_$1.err").read()[-600:])
PY
}
SO="--second_order 1 --seqs 1 --steps 12 --warmup 3"
one so_base "DYB_CONV_PAIR=0 DYB_HVP_GN_ONEPASS=0" "$SO"
one so_pair "DYB_CONV_PAIR=1 DYB_HVP_GN_ONEPASS=0" "$SO"
one so_gn1 "DYB_CONV_PAIR=0 DYB_HVP_GN_ONEPASS=1" "$SO"
one so_both "" "$SO"
one so_both_ov0 "DYB_HVP_OVERLAP=0" "$SO"
one so_full "" "--second_order 1 --full_losses 1 --inner_step 1 --seqs 1 --steps 6 --warmup 2"
one so_b16_base "DYB_CONV_PAIR=0 DYB_HVP_GN_ONEPASS=0" "--second_order 1 --batch 16 --seqs 1 --steps 6 --warmup 2"
one so_b16 "" "--second_order 1 --batch 16 --seqs 1 --steps 6 --warmup 2"
timeout 900 python -m pytest tests/test_adaptation_gpu.py -q -m gpu -k "second_order_inner3_exact_hvp or second_order_full_loss_set_matches or second_order_full_loss_set_vs_oracle" > $O/pytest_so.txt 2>&1; tail -3 $O/pytest_so.txt
DYB_BENCH_SMOKE_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --seqs 4 --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --percentile_frames 0 --replicas 1 > $O/bench_2rank.json 2> $O/bench_2rank.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_2rank.json").read().strip().splitlines()[-1])
    print("2-rank control flow:", d["n_gpus"], round(d["value"], 1), {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.items() if k in ("pw3d_operating_point", "scaling")})
except Exception as e:
    print("2-rank failed:", e, open("$O/bench_2rank.err").read()[-800:])
PY
