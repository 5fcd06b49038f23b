#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_cc
for v in diag_stamps diag_stamps_ahead; do echo "== $v"; AXVS_LIB_PATH=tools/$v.so timeout 300 python3 tools/r5/mt1_stamps.py 1,4,16,16 1,4,32,32 2>&1 | grep -v amdgpu; AXVS_LIB_PATH=tools/$v.so timeout 300 python3 tools/r5/cc_traj_stamps.py 2>&1 | grep -v amdgpu | head -8; done > gpurun_out/r5_cc/ahead_stamps.txt 2>&1
cat gpurun_out/r5_cc/ahead_stamps.txt
