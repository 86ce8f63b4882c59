"""A/B of library variants on the fluid20m bench: python tools/ab_fluid.py name=lib.so[,ENV=VAL] ..."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    name, rest = spec.split("=", 1)
    parts = rest.split(",")
    env = dict(os.environ)
    if parts[0]:
        env["SPARKL_HIP_LIB"] = os.path.join(root, parts[0])
    for kv in parts[1:]:
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--config", "fluid20m", "--steps", "50", "--warmup", "5"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", r.stderr[-400:]); continue
    o = json.loads(line[-1])
    print(f"{name:10s} ms/step {o['ms_per_step']:.4f} frac {o['step_frac_of_hbm_peak']:.3f} " + " ".join(f"{a}={b:.4f}" for a, b in o["kernels_avg_launch_ms"].items()), flush=True)
