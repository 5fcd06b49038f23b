#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run8; mkdir -p $O
AXVS_LIB_PATH=$R/tools/ab/hyb1.so timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -q -k "golden or cfg5 or merged_qkv_launch or ragged or shard or sweep" > $O/pytest_hyb1.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_hyb1.txt; tail -8 $O/pytest_hyb1.txt
bash tools/ab_run.sh 3 > $O/ab_metric.txt 2>&1
tail -22 $O/ab_metric.txt
