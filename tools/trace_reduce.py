#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace CSV to the launches of bench.py's TIMED region.

    python3 tools/trace_reduce.py <dir with *_kernel_trace.csv> <bench.log> [--steps K] [--out kernel_stats_timed.csv]

bench.py run with --no-extras --no-qkav --no-stages launches the layer's kernels only in its settling phase, its W warm-up steps and
its K timed steps, in that order, so the LAST K dispatches of every layer kernel ARE the timed region.  (Round 4's summaries were
`rocprofv3 --stats` averages over every launch of the process, including 500 launches per kernel that bench.py's QK^T/AV probe
truncates with option spatial_only: 52.7 us "average" for a kernel that takes 57.7.)  Per kernel: calls per step, mean / min / max /
stddev of the duration over the timed region; then the kernel sum per step, the period of the timed region in the trace
(first start to last end, / K) and the launch_us bench.py printed for the same run -- the three must agree.
"""
import argparse
import csv
import glob
import json
import math
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("bench_log", nargs="?")
    ap.add_argument("--steps", type=int, default=0, help="K of the bench run (default: read from the bench line)")
    ap.add_argument("--match", default="axvs::", help="substring of the kernel names that belong to a step")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    line = None
    if a.bench_log and os.path.exists(a.bench_log):
        for l in open(a.bench_log):
            if l.startswith("{"):
                line = json.loads(l)
    K = a.steps or (line["steps"] if line else 0)
    if K <= 0:
        sys.exit("number of timed steps unknown: pass --steps")
    files = sorted(glob.glob(os.path.join(a.trace_dir, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit(f"no *kernel_trace.csv under {a.trace_dir}")
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if a.match not in n or "pack" in n or "pos3d" in n:
                continue
            per.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not per:
        sys.exit("no matching kernels in the trace")
    for v in per.values():
        v.sort()
    least = min(len(v) for v in per.values() if len(v) >= K)      # the least-launched layer kernel runs once per step
    rows, ksum, t_first, t_last = [], 0.0, None, None
    for n, v in sorted(per.items(), key=lambda kv: -len(kv[1])):
        if len(v) < K:
            continue
        cps = round(len(v) / least)
        reg = v[-K * cps:]
        d = [e - s for s, e in reg]
        mean = sum(d) / len(d)
        sd = math.sqrt(sum((x - mean) ** 2 for x in d) / len(d))
        rows.append({"Name": n, "Calls": len(d), "CallsPerStep": cps, "AverageNs": round(mean, 1), "MinNs": min(d), "MaxNs": max(d),
                     "StdDevNs": round(sd, 1), "CallsInWholeTrace": len(v)})
        ksum += mean * cps
        t_first = reg[0][0] if t_first is None else min(t_first, reg[0][0])
        t_last = reg[-1][1] if t_last is None else max(t_last, reg[-1][1])
    period = (t_last - t_first) / K
    out = a.out or os.path.join(a.trace_dir, "kernel_stats_timed.csv")
    with open(out, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            w.writerow(r)
        fh.write(f"# timed region = the last {K} step(s) of the trace; kernel sum per step {ksum / 1e3:.2f} us; "
                 f"period of the region (first start .. last end) / K {period / 1e3:.2f} us")
        if line:
            lu = line.get("roofline", {}).get("launch_us")
            fh.write(f"; bench.py of the same run: ms_per_step {line['ms_per_step'] * 1e3:.2f} us, roofline.launch_us {lu}")
        fh.write("\n")
    for r in rows:
        print(f"  {r['Name'].replace('void axvs::', '')[:78]:78s} x{r['CallsPerStep']}  avg {r['AverageNs'] / 1e3:8.2f} us  "
              f"(min {r['MinNs'] / 1e3:.2f}, max {r['MaxNs'] / 1e3:.2f}, sd {r['StdDevNs'] / 1e3:.2f})")
    print(f"  kernel sum per step {ksum / 1e3:.2f} us | trace period per step {period / 1e3:.2f} us"
          + (f" | bench launch_us {line.get('roofline', {}).get('launch_us')} ms_per_step {line['ms_per_step'] * 1e3:.2f} us" if line else ""))
    print(f"  -> {out}")


if __name__ == "__main__":
    main()
