#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 1200 python3 tools/r5/merge_small_ab.py > gpurun_out/r5_plan/merge_small.txt 2>&1; tail -20 gpurun_out/r5_plan/merge_small.txt
