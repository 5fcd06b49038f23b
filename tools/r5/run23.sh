#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_cc
AXVS_LIB_PATH=tools/diag_stamps.so timeout 300 python3 tools/r5/mt1_stamps.py 1,4,16,16 1,4,32,32 1,4,24,40 1,4,16,64 > gpurun_out/r5_cc/mt1_stamps.txt 2>&1; tail -20 gpurun_out/r5_cc/mt1_stamps.txt
