"""Run bench.py with the given args and print a compact summary (tuning helper)."""
import json, subprocess, sys
p = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in p.stdout.splitlines() if l.startswith("{")]
if not line:
    print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
d = json.loads(line[-1]); r = d["roofline"]
print(f"value {d['value']} frames/s  ms/step {d['ms_per_step']}  fwd_us {r['launch_us']}  frac {r['frac']}")
print("  ".join(f"{k}={v}" for k, v in r["stage_us"].items()))
if "cpu_baseline" in d: print(d["cpu_baseline"])
