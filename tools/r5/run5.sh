#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run5; mkdir -p $O
for v in st_ws2 st_ws0p st_old; do
  echo "== $v" >> $O/stamps.txt
  AXVS_LIB_PATH=$R/tools/ab/$v.so python3 tools/r5/ffn_stamps.py 13 >> $O/stamps.txt 2>&1
done
cat $O/stamps.txt
mkdir -p $R/tools/abkeep; mv $R/tools/ab/st_*.so $R/tools/abkeep/
bash tools/ab_run.sh 2 > $O/ab_metric.txt 2>&1
tail -22 $O/ab_metric.txt
