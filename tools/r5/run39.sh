#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 1500 python3 tools/sweep_shapes.py 160 11 > gpurun_out/r5_plan/sweep_random_11.txt 2>&1; grep -c "repeat-equal True" gpurun_out/r5_plan/sweep_random_11.txt; grep -n "FAIL\|worst\|Error\|error" gpurun_out/r5_plan/sweep_random_11.txt | head
timeout 1500 python3 tools/sweep_shapes.py 160 12 > gpurun_out/r5_plan/sweep_random_12.txt 2>&1; grep -c "repeat-equal True" gpurun_out/r5_plan/sweep_random_12.txt; grep -n "FAIL\|worst\|Error\|error" gpurun_out/r5_plan/sweep_random_12.txt | head
