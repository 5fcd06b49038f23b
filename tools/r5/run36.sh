#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 900 python3 tools/r5/opt_sweep.py qkv_split_upto 64,85,128,100000 2,5,24,40 1,8,24,40 1,7,25,43 1,5,40,40 1,5,48,48 1,3,17,127 1,4,17,127 1,5,64,64 2>&1 | grep -v amdgpu > gpurun_out/r5_plan/qkv_split.txt
cat gpurun_out/r5_plan/qkv_split.txt
