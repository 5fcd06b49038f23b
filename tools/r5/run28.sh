#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python3 tools/r5/merge_mid_dbg.py 2>&1 | grep -v amdgpu 
