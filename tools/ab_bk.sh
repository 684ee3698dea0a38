mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 120 python tools/enginebench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fwd %.3f  bwd_aux %.3f  bwd_1s %.3f' % (d['forward']['gpu_ms'], d['backward_aux']['gpu_ms'], d['backward_1stream']['gpu_ms']))"; }
run A=bk32
cp dynaboa_amd/libdynaboa_hip.so /tmp/lib32.so; cp dynaboa_amd/libdynaboa_hip_bk64.so dynaboa_amd/libdynaboa_hip.so
run A=bk64
run A=bk64 DYB_KSTEP_US=1.0
run A=bk64 DYB_KSTEP_US=1.0 DYB_MIN_STEPS=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv or bottleneck or engine" 2>&1 | tail -2
timeout 200 python bench.py --steps 40 --warmup 8 --no_cpu_baseline --no_roofline 2>/dev/null | cut -c1-200
cp /tmp/lib32.so dynaboa_amd/libdynaboa_hip.so
