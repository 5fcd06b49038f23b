#!/bin/bash
# cross-clip module (BASELINE config 4) with every tools/ab/*.so, interleaved, on this box:  tools/cc_ab.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in $(seq 1 ${1:-2}); do
  for so in $R/tools/ab/*.so; do
    echo "$(basename $so .so): $(AXVS_LIB_PATH=$so python3 $R/tools/cc_time.py 2>/dev/null | head -1 | sed 's/.*64x64): //; s/ ->.*//')"
  done
done
