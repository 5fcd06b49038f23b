"""Scan hipcc's gfx950 ISA for code paths that can have >= LIMIT LGKM-counted operations (LDS + scalar-memory) in flight
before an `s_waitcnt lgkmcnt(..)`.  lgkmcnt is a 4-bit counter: 16 outstanding operations wrap it (see lds_fence() in
axvs_common.h).  Linear scan per kernel (branch targets do not reset the count: conservative).

    python tools/check_lgkm.py [file.s ...]     # with no argument: compiles axvs_api.hip to ISA first
"""
import os, re, subprocess, sys, tempfile

LIMIT = 14
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path):
    worst = {}
    kernel, n, peak, where = None, 0, 0, 0
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel, n, peak = m.group(1), 0, 0
            continue
        if kernel is None or not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if op == "s_endpgm":
            worst[kernel] = (peak, where)
            kernel = None
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = min(n, int(m.group(1)))
        elif op.startswith(("ds_", "s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            n += 1
            if n > peak:
                peak, where = n, ln
    return worst


def main():
    files = sys.argv[1:]
    if not files:
        tmp = tempfile.mkdtemp()
        out = os.path.join(tmp, "axvs.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                               os.path.join(ROOT, "axial_vs_amd", "csrc", "axvs_api.hip")], stderr=subprocess.DEVNULL)
        files = [out]
    bad = 0
    for f in files:
        for k, (peak, where) in sorted(scan(f).items(), key=lambda kv: -kv[1][0]):
            flag = "  <-- TOO MANY" if peak >= LIMIT else ""
            if peak >= 8 or flag:
                print(f"{peak:3d} LGKM ops in flight  line {where:6d}  {k[:90]}{flag}")
            bad += peak >= LIMIT
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
