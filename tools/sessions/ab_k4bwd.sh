run() { echo "== $*"; env "$@" timeout 100 python tools/enginebench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fwd %.3f  bwd_aux %.3f (host %.3f)  bwd_1s %.3f' % (d['forward']['gpu_ms'], d['backward_aux']['gpu_ms'], d['backward_aux']['host_issue_ms'], d['backward_1stream']['gpu_ms']))"; }
run DYB_K4_BWD=0
run DYB_K4_BWD=1
timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_adaptation_gpu.py -m gpu -q -p no:cacheprovider -k "dgrad_gn_reduce or k4_backward" 2>&1 | tail -2
