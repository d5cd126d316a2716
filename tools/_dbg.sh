#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4l; mkdir -p $O
: > $O/mesh41_per.txt
for i in 1 2 3; do CHECK_NO_DOWNLOAD=1 CHECK_IDLE_S=0.4 FEMFCT_DEBUG_TIMES=20 timeout -k 10 100 python3 tools/mesh_step_check.py 40 50 64,32,128,64 2>&1 | grep "N=41\|took" | cut -c1-100; done >> $O/mesh41_per.txt
