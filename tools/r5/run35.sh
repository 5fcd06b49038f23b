#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
{ timeout 600 python3 tools/r5/stack_ab.py no_merge_qkv 1,4,64,64 2,4,64,64 1,2,49,85 1,4,48,80 1,4,32,64
timeout 600 python3 tools/r5/stack_ab.py no_ffn_fusion 1,4,64,64 1,2,49,85 1,4,32,64 1,2,48,80; } 2>&1 | grep -v amdgpu > gpurun_out/r5_plan/stack_ab_main.txt
cat gpurun_out/r5_plan/stack_ab_main.txt
