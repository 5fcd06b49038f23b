#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 900 python3 tools/r5/plan_sweep.py 1,4,32,34 1,4,32,36 1,4,36,36 1,2,32,68 1,2,40,56 1,2,36,64 1,5,12,20 1,5,24,40 1,5,32,32 1,5,40,40 1,5,48,48 1,5,64,64 1,6,32,32 1,6,48,48 1,8,24,40 1,8,32,32 1,8,48,48 2,5,24,40 > gpurun_out/r5_plan/sweep2.txt 2>&1
tail -40 gpurun_out/r5_plan/sweep2.txt
