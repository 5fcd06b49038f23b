#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_cc
timeout 900 python3 tools/r5/few_rows_ab.py > gpurun_out/r5_cc/few_rows_ab.txt 2>&1; tail -14 gpurun_out/r5_cc/few_rows_ab.txt
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged or small_problem or cross_clip or fused or tube or golden" 2>&1 | tail -3
