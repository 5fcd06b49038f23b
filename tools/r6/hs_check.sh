#!/bin/bash
# head-split tier: parity subset + cross-clip module time.  tools/r6/hs_check.sh <tag> [pytest -k expression]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
K=${2:-"cross_clip_module_golden or shipped_map_sizes or axial_layer_golden or encoder_golden"}
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -25 > gpurun_out/hs_$1_tests.txt
cat gpurun_out/hs_$1_tests.txt
for o in "hsplit=1" "hsplit=0"; do
  timeout 300 python bench.py --workload cc --no-cpu-baseline --opt $o 2>/dev/null | tail -1 | python -c "import sys,json; print('$o', round(json.loads(sys.stdin.read())['ms_per_step']*1000,1), 'us')"
done | tee gpurun_out/hs_$1_bench.txt
