#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s31; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_adaptation_gpu.py -q -x -k "fused_level_node_matches" 2>&1 | tail -8
