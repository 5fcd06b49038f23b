#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run6; mkdir -p $O; rm -f $O/kstamps.txt
for v in 0 1 2; do
  echo "== chunk $v" >> $O/kstamps.txt
  FFNK=$v AXVS_LIB_PATH=$R/tools/ab/kst$v.so python3 tools/r5/ffnk_stamps.py >> $O/kstamps.txt 2>&1
done
cat $O/kstamps.txt
