#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 0 1; do echo "== conv_nt128_exact=$o"; timeout 600 python3 tools/stack_precision.py conv_nt128_exact=$o 2>&1 | grep -v amdgpu; done
echo "== conv_nt128_exact=0 msda_gemm=2"; timeout 600 python3 tools/stack_precision.py conv_nt128_exact=0 msda_gemm=2 2>&1 | grep -v amdgpu
