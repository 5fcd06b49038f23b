#!/bin/bash
# stand-alone FFN kernel (axvs_ffn_fwd, M rows) for every tools/ab/*.so, interleaved on this box: tools/ffn_ab.sh [M]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
M=${1:-16384}
for r in 1 2 3; do
  for so in $R/tools/ab/*.so; do
    echo -n "$(basename $so .so) M=$M: "; AXVS_LIB_PATH=$so python3 $R/tools/ffn_time.py $M 2>&1 | tail -1
  done
done
