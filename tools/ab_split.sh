mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 120 python tools/enginebench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fwd %.3f  bwd_aux %.3f  bwd_1s %.3f' % (d['forward']['gpu_ms'], d['backward_aux']['gpu_ms'], d['backward_1stream']['gpu_ms']))"; }
F=DYB_FOLD_IN_REDUCE=1
run $F
run $F DYB_RAW_FOLD_US=0.2 DYB_RAW_SLAB_US=0.08
run $F DYB_RAW_FOLD_US=0.2 DYB_RAW_SLAB_US=0.04
run $F DYB_KSTEP_US=1.0
run $F DYB_KSTEP_US=2.0
run $F DYB_KSTEP_US=1.0 DYB_GRID_CAP=2048
run $F DYB_KSTEP_US=1.0 DYB_MIN_STEPS=1
run $F DYB_KSTEP_US=1.0 DYB_MIN_STEPS=1 DYB_GRID_CAP=2048
run $F DYB_KSTEP_US=1.0 DYB_GRID_CAP=512
run $F DYB_FWD_SLAB_US=0.05
run $F DYB_FWD_SLAB_US=0.4
