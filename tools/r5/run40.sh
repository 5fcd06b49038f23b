#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
for o in "" "--opt merge_qkv_any=1" "--opt merge_qkv_any=1 --opt merge_stagger_us=25" "--opt merge_qkv_any=1 --opt merge_stagger_us=50" "--opt merge_qkv_any=1 --opt merge_stagger_us=75" "" "--opt merge_qkv_any=1" "--opt merge_qkv_any=1 --opt merge_stagger_us=50"; do
timeout 300 python3 bench.py --shape 8,4,256,96,96 --steps 40 --no-cpu-baseline --no-extras --no-qkav $o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('cfg5 share [$o]', round(j['ms_per_step']*1e3,1), j['roofline'].get('stage_us'))
"; done > gpurun_out/r5_plan/stagger.txt 2>&1
cat gpurun_out/r5_plan/stagger.txt
