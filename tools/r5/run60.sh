#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 1500 python3 tools/sweep_shapes.py 200 21 > gpurun_out/r5_plan/sweep_random_21.txt 2>&1; grep -c "repeat-equal True" gpurun_out/r5_plan/sweep_random_21.txt; grep -n "FAIL\|worst\|Error\|error" gpurun_out/r5_plan/sweep_random_21.txt | head
