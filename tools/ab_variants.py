"""Build A/B variants of libaxvs.so (tools/ab/<name>.so, untracked) for one-box comparisons with tools/ab_run.sh:
    python tools/ab_variants.py name1=-DFLAG=0,-DOTHER=1 name2= ...
A variant with no flags is the shipped code.  Kernel timings of different GPU boxes (or of different processes' clocks) are
not comparable; these builds run back to back on one box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
os.makedirs(os.path.join(ROOT, "tools", "ab"), exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    fl = [f for f in flags.split(",") if f]
    ge.build(force=True, extra_flags=fl, lib_path=os.path.join(ROOT, "tools", "ab", name + ".so"), load=False)
    print("built", name, fl)
