#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in occ1 occ2w; do
AXVS_LIB_PATH=tools/ab/$v.so timeout 300 python3 bench.py --shape 8,4,256,96,96 --steps 40 --no-cpu-baseline --no-extras --no-qkav 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$v cfg5 share', round(j['ms_per_step']*1e3,1), j['roofline'].get('stage_us'))
"; done; done
AXVS_LIB_PATH=tools/ab/occ2w.so timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged or golden or small_problem or fused_qkv or two_streams" 2>&1 | tail -3
