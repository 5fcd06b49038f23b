#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 600 python3 tools/r5/vip_module_time.py > gpurun_out/r5_plan/vip_module.txt 2>&1; tail -8 gpurun_out/r5_plan/vip_module.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5_plan/vipprof -o vip -- python3 $GRAFT_REPO_ROOT/tools/r5/vip_module_time.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/r5_plan/vipprof/*kernel_stats.csv gpurun_out/r5_plan/vipprof/*/*kernel_stats.csv 2>/dev/null | head -1); head -25 $f | cut -c1-200
