#!/bin/bash
# r05 s8: is one sequence bound by the host's launch rate?  host cost of issuing one frame into an empty queue vs its device time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/s8; O=gpurun_out/s8; export TMPDIR=/tmp
timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee $O/host_floor.txt
DYB_SIDE_THREAD=1 timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee -a $O/host_floor.txt
DYB_NO_AUX=1 timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee -a $O/host_floor.txt
HIP_FORCE_DEV_KERNARG=1 timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee -a $O/host_floor.txt
HIP_FORCE_DEV_KERNARG=0 timeout 200 python tools/host_floor.py 40 2>&1 | tail -1 | tee -a $O/host_floor.txt
