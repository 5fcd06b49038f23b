#!/bin/bash
# cross-clip module: parity tests + module time + per-kernel times.  tools/r6/cc_check.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "cross_clip or tube_link_head or tube_link_cross" 2>&1 | tail -5 > gpurun_out/cc_$1_tests.txt
cat gpurun_out/cc_$1_tests.txt
for i in 1 2; do timeout 300 python bench.py --workload cc --no-cpu-baseline 2>/dev/null | tail -1; done | tee gpurun_out/cc_$1_bench.txt
bash tools/cc_prof.sh $1 2>&1 | tail -30 | tee gpurun_out/cc_$1_kernels.txt
