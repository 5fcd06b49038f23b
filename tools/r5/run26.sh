#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
for sh in 1,5,256,24,40 1,5,256,32,32 1,6,256,32,32 1,8,256,24,40 1,5,256,40,40 2,5,256,24,40; do for o in 65 128 200 65 128 200; do
timeout 300 python3 bench.py --shape $sh --no-cpu-baseline --no-extras --no-qkav --steps 400 --opt ffn_split_below=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$sh ffn_split_below=$o', round(j['ms_per_step']*1e3,2), j['roofline'].get('stage_us'))
"; done; done > gpurun_out/r5_plan/ffn_split.txt 2>&1
cat gpurun_out/r5_plan/ffn_split.txt
