#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 1500 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged or two_streams or graphed or hand_off or soak or merged" 2>&1 | tail -4
timeout 600 python3 tools/r5/merge_mid_ab.py 1,5,24,40 1,5,32,32 2,5,24,40 1,8,32,32 1,8,24,40 > gpurun_out/r5_plan/merge_mid_default.txt 2>&1; tail -6 gpurun_out/r5_plan/merge_mid_default.txt
timeout 600 python3 tools/merge_soak.py > gpurun_out/r5_plan/soak.txt 2>&1; tail -4 gpurun_out/r5_plan/soak.txt
