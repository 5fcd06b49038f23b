#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "cross_clip or tube_link" 2>&1 | tail -6
