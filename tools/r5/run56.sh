#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in rowmajor headmajor; do
AXVS_LIB_PATH=tools/ab/$v.so timeout 600 python3 bench.py --no-cpu-baseline --no-qkav 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('$v: cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'])
"; done; done
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "msda or deform or within_clip or decoder or tube_link_plugin or plugin" 2>&1 | tail -3
