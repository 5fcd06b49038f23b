#!/bin/bash
# HBM traffic (tools/pmc_traffic.sh) of every tools/ab/*.so on this box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for so in $R/tools/ab/*.so; do
  echo "== $(basename $so .so)"
  AXVS_LIB_PATH=$so bash $R/tools/pmc_traffic.sh 2>&1 | grep -E "read_MB_per_launch|write_MB_per_launch|layer_"
  cp $R/gpurun_out/pmc_traffic/traffic.json $R/gpurun_out/traffic_$(basename $so .so).json
  rm -rf $R/gpurun_out/pmc_traffic/*/
done
