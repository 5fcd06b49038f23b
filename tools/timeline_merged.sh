#!/bin/bash
# phase timelines of the merged q/k/v + trajectory kernels (-DAXVS_STAMPS build: tools/diag_stamps.so) into gpurun_out/timeline_merged.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
J=$R/gpurun_out/timeline_merged.json
mkdir -p $R/gpurun_out; rm -f $J
ORDW=0,16,17,18,19,20,21,22,23,11,12,13,14,15,1,2,3,4,5,9,6,7,8,10
ORDH=32,48,49,50,51,52,53,54,55,43,44,45,46,47,33,34,35,36,37,41,38,39,40,42
AXVS_STAMPS_JSON=$J AXVS_STAMPS_TAG="temporal_fused_kernel<f16,T=4,MT=4,NKS=2,FFN,MQ=2> (merged width pass + FFN)" AXVS_LIB_PATH=$R/tools/diag_stamps.so python3 $R/tools/stamps.py 24 $ORDW | tail -26
AXVS_STAMPS_JSON=$J AXVS_STAMPS_TAG="temporal_fused_kernel<f16,T=4,MT=4,NKS=2,MQ=2> (merged height pass)" AXVS_LIB_PATH=$R/tools/diag_stamps.so python3 $R/tools/stamps.py 24 $ORDH | tail -26
