#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python3 tools/r5/conv_exact.py 2>&1 | grep "splitk="
timeout 1500 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "conv1x1 or within_clip or decoder or pixel" 2>&1 | tail -4
for o in 0 1 0 1; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt conv_nt128_splitk=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('conv_nt128_splitk=$o: cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'])
"; done
