"""Brief of a bench.py JSON line (file argument): the numbers the round's notes quote."""
import json, sys
o = json.load(open(sys.argv[1]))
r = o.get("roofline") or {}
print("ms/step", round(o["ms_per_step"], 4), "value %.4g" % o["value"], "frac", round(r.get("frac", 0), 3), "kernel_ms", round(r.get("avg_launch_ms", 0), 4),
      "cold", round((o.get("cold_window") or {}).get("ms_per_step", 0), 4), "late", round((o.get("late_collapse") or {}).get("ms_per_step", 0), 4),
      "deep", round((o.get("deep_collapse") or {}).get("ms_per_step", 0), 4))
if o.get("strong_projection"):
    print("strong_projection", {k: v for k, v in o["strong_projection"].items() if k not in ("what", "not_included")})
if o.get("cpu_baseline"):
    c = o["cpu_baseline"]; print("cpu", "%.3g" % c["value"], "cores", c["cores"], {k: "%.3g" % v for k, v in c["thread_probe_particle_steps_per_s"].items()})
if o.get("weak_scaling"):
    print("weak", {k: v for k, v in o["weak_scaling"].items() if k != "note"})
