#!/bin/bash
# kernel stats of the layer at one shape for every tools/ab/*.so: tools/shape_ab.sh B T H W [option=value]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in 1 2; do for so in $R/tools/ab/*.so; do echo "== $(basename $so .so)"; AXVS_LIB_PATH=$so bash $R/tools/shape_prof.sh "$@" 2>&1 | grep -v amdgpu | tail -4; done; done
