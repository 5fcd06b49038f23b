#!/bin/bash
# timeline of ONE forward of the within-clip module at BASELINE config 3 (kernel starts / ends / queue, gaps between
# dependent launches): tools/cfg3_trace.sh [options...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/cfg3trace
rm -rf $OUT; mkdir -p $OUT
AXVS_CFG3_NO_GRAPH=${AXVS_CFG3_NO_GRAPH-1} rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/cfg3_time.py 40 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a forward starts with the first pos-free kernel after the longest idle; use the gemm of the input projection as marker
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "msda_gather_kernel" in n]
per = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
# take the 10th last full period (steady state)
a, b = marks[-12 - (2 if "AXVS_TRACE_FWD" in __import__("os").environ else 0)], marks[-11] if "AXVS_TRACE_FWD" not in __import__("os").environ else marks[-12]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
busy_end = 0
lines = []
qs = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.setdefault(r["Queue_Id"], len(qs))
    gap = s - busy_end
    busy_end = max(busy_end, e)
    nm = r["Kernel_Name"].replace("void ", "").replace("axvs::", "")[:58]
    lines.append(f"{s/1e3:9.2f} {e/1e3:9.2f} q{q} dur {(e-s)/1e3:7.2f} idle-before {gap/1e3:7.2f}  {nm}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
print("\n".join(lines))
tot = int(rows[b]["Start_Timestamp"]) - t0
ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
u, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"period {tot/1e3:.1f} us, launches {len(seg)}, kernel sum {ksum/1e3:.1f} us, GPU busy (union) {u/1e3:.1f} us, idle {(tot-u)/1e3:.1f} us")
PY
cp $OUT/log.txt $OUT/../cfg3trace_log.txt 2>/dev/null
rm -rf $OUT/*/
