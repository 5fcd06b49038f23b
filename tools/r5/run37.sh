#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "two_chunks or small_problem or ragged" 2>&1 | tail -4
timeout 900 python3 tools/r5/opt_sweep.py ffn_split_pairs 1,0 1,5,24,40 1,5,32,32 1,6,32,32 1,5,25,43 1,8,32,32 1,5,40,40 1,7,25,43 1,8,24,40 2>&1 | grep -v amdgpu > gpurun_out/r5_plan/ffn_pairs.txt
cat gpurun_out/r5_plan/ffn_pairs.txt
