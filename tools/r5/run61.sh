#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "in_launch_finishing" 2>&1 | grep -v "^$" | tail -25
