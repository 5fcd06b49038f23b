#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r5_plan/gpu_tests3.txt 2>&1; tail -4 gpurun_out/r5_plan/gpu_tests3.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
