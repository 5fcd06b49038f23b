#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "two_chunks or small_problem or ragged or gelu" 2>&1 | tail -3
