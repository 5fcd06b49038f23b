#!/bin/bash
# phase timelines + weight-stream ablation into gpurun_out/timeline.json (diagnostic builds: tools/diag_*.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
J=$R/gpurun_out/timeline.json
rm -f $J
ORD=0,11,12,13,14,15,1,2,3,4,5,9,6,7,8,10
AXVS_STAMPS_JSON=$J AXVS_STAMPS_TAG="temporal_fused_kernel<f16,T=4,MT=4,NKS=2,FFN> (width pass + FFN)" AXVS_LIB_PATH=$R/tools/diag_stamps.so python3 $R/tools/stamps.py 16 $ORD | tail -18
AXVS_STAMPS_JSON=$J AXVS_STAMPS_TAG="qkv_fused_kernel<f16> (width pass, positions generated)" AXVS_LIB_PATH=$R/tools/diag_stamps_qkv.so python3 $R/tools/stamps.py 9 | tail -10
for l in libaxvs diag_ablw; do f=$R/tools/$l.so; [ $l = libaxvs ] && f=$R/axial_vs_amd/libaxvs.so
  AXVS_LIB_PATH=$f python3 $R/bench.py --steps 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$l', j['ms_per_step'], j['roofline']['launch_us'], j['roofline']['stage_us'])
        p = '$J'; d = json.load(open(p)); d.setdefault('weight_stream_ablation', {})['$l'] = {'ms_per_step': j['ms_per_step'], 'launch_us': j['roofline']['launch_us'], 'stage_us': j['roofline']['stage_us']}
        d['weight_stream_ablation']['note'] = 'diag_ablw = -DAXVS_ABL_W: every weight-fragment load hits one L1-resident KiB (results wrong on purpose): the difference is the exposed L2 -> CU weight stream'
        json.dump(d, open(p, 'w'), indent=1)
"; done
