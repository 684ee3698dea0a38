run() { echo "== $*"; env "$@" timeout 120 python tools/enginebench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fwd %.3f (host %.3f)  bwd_aux %.3f' % (d['forward']['gpu_ms'], d['forward']['host_issue_ms'], d['backward_aux']['gpu_ms']))"; }
run DYB_K4=0
run DYB_K4=1 DYB_K4_MAXC=256
run DYB_K4=1 DYB_K4_MAXC=512
run DYB_K4=1 DYB_K4_MAXC=1024
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "layer_gnstats or engine" 2>&1 | tail -1
