#!/bin/bash
# planner thresholds: where do the few-rows forms stop paying?
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 900 python3 tools/r5/plan_sweep.py > gpurun_out/r5_plan/sweep.txt 2>&1
tail -40 gpurun_out/r5_plan/sweep.txt
