#!/bin/bash
# Register / spill summary of the fused trajectory kernels of one instantiation unit:
#   tools/kernel_resources.sh <bf 0|1> <T> [extra hipcc flags]
# prints one line per kernel: template arguments <BF,T,MT,NKS,FFN,VROW,QKVN,MQ>, VGPRs, spilled VGPRs, scratch bytes per lane.
R=$(cd "$(dirname "$0")/.." && pwd)
BF=${1:-0}; T=${2:-4}; shift 2
cd "$R/axial_vs_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DAXVS_INST_BF=$BF -DAXVS_INST_T=$T "$@" \
  -Rpass-analysis=kernel-resource-usage -o /dev/null axvs_temporal_inst.hip 2>&1 |
python3 -c '
import re, sys
name = None
vals = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); vals = {}
        continue
    for key in ("VGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
        if m: vals[key] = int(m.group(1))
    if "LDS Size" in line and name:
        t = re.search(r"temporal_fused_kernelI(.*?)EEv", name)
        if t:
            args = re.findall(r"L[bi](\d+)E", t.group(1) + "E")
            print("<%s>" % ",".join(args), "vgpr", vals.get("VGPRs"), "vspill", vals.get("VGPRs Spill"), "sspill", vals.get("SGPRs Spill"),
                  "scratch", vals.get("ScratchSize [bytes/lane]"))
        name = None
'
