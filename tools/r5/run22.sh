#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_cc
AXVS_LIB_PATH=tools/diag_stamps.so timeout 300 python3 tools/r5/cc_traj_stamps.py > gpurun_out/r5_cc/traj_stamps.txt 2>&1; tail -20 gpurun_out/r5_cc/traj_stamps.txt
