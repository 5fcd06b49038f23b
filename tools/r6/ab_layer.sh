#!/bin/bash
# interleaved A/B of tools/ab/*.so on the headline bench line (no profiler): ms_per_step, qk_av_frac.  tools/r6/ab_layer.sh [rounds] [bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-3}; shift
for r in $(seq 1 $R); do
  for so in tools/ab/*.so; do
    n=$(basename $so .so)
    AXVS_LIB_PATH=$PWD/$so python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); rf = d['roofline']
print('$n', 'us/step', round(d['ms_per_step']*1e3, 2), 'launch_us', rf['launch_us'], 'qk_av', rf.get('qk_av_frac'), 'stages', rf.get('stage_us'))"
  done
done | tee gpurun_out/ab_layer.txt
