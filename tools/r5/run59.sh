#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 600 python3 tools/r5/opt_sweep.py ffn_split_finish 1,0 1,4,16,16 1,4,32,32 1,2,25,43 1,5,12,20 1,5,24,40 2>&1 | grep -v amdgpu | cut -c1-110 | tee gpurun_out/r5_plan/ffn_fin2.txt
