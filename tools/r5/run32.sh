#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
{ timeout 600 python3 tools/r5/stack_ab.py merge_small 1,4,16,16 1,4,32,32 1,2,25,43 1,4,8,8
timeout 600 python3 tools/r5/stack_ab.py merge_mid 1,5,24,40 1,5,32,32 1,8,32,32 1,6,32,32
timeout 600 python3 tools/r5/stack_ab.py small_tiles_below 1,2,48,80 1,4,40,40 1,5,24,40; } 2>&1 | grep -v amdgpu > gpurun_out/r5_plan/stack_ab.txt
cat gpurun_out/r5_plan/stack_ab.txt
