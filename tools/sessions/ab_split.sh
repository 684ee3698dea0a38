mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 120 python tools/enginebench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fwd %.3f  bwd_aux %.3f  bwd_1s %.3f' % (d['forward']['gpu_ms'], d['backward_aux']['gpu_ms'], d['backward_1stream']['gpu_ms']))"; }
run A=1
run DYB_KSTEP_US=1.0
run DYB_KSTEP_US=1.5
run DYB_KSTEP_US=2.5
run DYB_KSTEP_US=1.5 DYB_FWD_SLAB_US=0.08
run DYB_KSTEP_US=1.5 DYB_FWD_SLAB_US=0.3
run DYB_KSTEP_US=1.5 DYB_GRID_CAP=2048
run DYB_KSTEP_US=1.5 DYB_RAW_SLAB_US=0.1
run DYB_KSTEP_US=1.5 DYB_RAW_SLAB_US=0.02
run DYB_KSTEP_US=4 DYB_MIN_STEPS=1 DYB_GRID_CAP=2048
