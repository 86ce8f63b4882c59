"""A/B of library variants on the bench workload: python tools/ab_bench.py [--steps K] name=lib.so[,ENV=VAL...] ...
Prints one line per variant: ms/step, dominant-kernel ms, roofline frac, late-phase numbers."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
extra = [a for a in sys.argv[1:] if a not in args]
for spec in args:
    name, rest = spec.split("=", 1)
    parts = rest.split(",")
    env = dict(os.environ)
    if parts[0]:
        env["SPARKL_HIP_LIB"] = os.path.join(root, parts[0])
    for kv in parts[1:]:
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", r.stderr[-400:])
        continue
    o = json.loads(line[-1])
    k = o.get("kernels_avg_launch_ms", {})
    late = o.get("late_collapse") or {}
    lk = late.get("kernels_avg_launch_ms", {})
    print(f"{name:14s} ms/step {o['ms_per_step']:.4f}  " + " ".join(f"{a}={b:.4f}" for a, b in k.items()) +
          f"  frac {o.get('roofline', {}).get('frac', 0):.3f} | late ms/step {late.get('ms_per_step', 0):.4f} " + " ".join(f"{a}={b:.4f}" for a, b in lk.items()) +
          f" late_frac {(late.get('roofline') or {}).get('frac', 0):.3f} resorts {late.get('resorts')} reorder_ms {late.get('reorder_ms_total', 0):.2f}" +
          (f" | deep ms/step {o['deep_collapse']['ms_per_step']:.4f} g2p2g={o['deep_collapse']['kernels_avg_launch_ms'].get('g2p2g', 0):.4f} frac {o['deep_collapse']['roofline']['frac']:.3f}" if o.get("deep_collapse") else ""), flush=True)
