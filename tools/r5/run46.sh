#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 128 64 128 64; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt gemm_small_upto=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('gemm_small_upto=$o (x2 for K >= 1024): cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'], 'cc', e['cc_cfg4']['us_per_forward'], 'train', e['train_step']['ms_per_step'], 'cc_train', e['cc_train_cfg4']['ms_per_step'], 'wc_train', e['wc_train_cfg3']['ms_per_step'])
"; done
timeout 1200 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "within_clip or decoder or conv1x1 or msda or cross_clip" 2>&1 | tail -3
