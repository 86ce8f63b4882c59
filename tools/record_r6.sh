#!/bin/bash
# record_r6.sh <bench|prof|tests>: the commands behind profiles/r06_bench_*.json, profiles/r06_*_kernel_stats.csv / _pmc_summary.txt and
# profiles/r06_parity.jsonl (GPU box: gpurun -- bash tools/record_r6.sh <case>; outputs under gpurun_out/r06/, copied into profiles/ by hand).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
line() { grep '^{' | tail -1; }
case "$1" in
bench)
python bench.py                                        2> gpurun_out/r06/sand.err          | line > gpurun_out/r06/bench_sand.json
python bench.py --steps 20 --warmup 5                  2> gpurun_out/r06/sand_driver.err   | line > gpurun_out/r06/bench_sand_driver.json
python bench.py --config fluid20m                      2> gpurun_out/r06/fluid.err         | line > gpurun_out/r06/bench_fluid.json
python bench.py --config fluid20m --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null     | line > gpurun_out/r06/bench_fluid_driver.json
python bench.py --config neo1m                         2> gpurun_out/r06/neo.err           | line > gpurun_out/r06/bench_neo.json
python bench.py --config neo1m --steps 20 --warmup 5 --no-cpu-baseline    2> /dev/null     | line > gpurun_out/r06/bench_neo_driver.json
python bench.py --config snow1m                        2> gpurun_out/r06/snow.err          | line > gpurun_out/r06/bench_snow.json
python bench.py --config sandblock10m --no-cpu-baseline 2> gpurun_out/r06/sandblock.err    | line > gpurun_out/r06/bench_sandblock.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/bench_*.json")):
    try: o = json.loads(open(f).read())
    except Exception as e: print(f, "UNREADABLE", e); continue
    r = o.get("roofline", {}); c = o.get("cold_window") or {}; l = o.get("late_collapse") or {}; d = o.get("deep_collapse") or {}
    if o.get("strong_projection"): print("  strong_projection", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in o["strong_projection"].items() if k in ("group_ms_per_substep_one_device", "per_slab_ms_per_substep", "fixed_cost_us_per_slab_substep", "projected_speedup_at_8_gpus")})
    print(f.split("/")[-1], "ms", round(o["ms_per_step"], 4), "value", "%.4g" % o["value"], "frac", round(r.get("frac", 0), 3), "kernel_ms", round(r.get("avg_launch_ms", 0), 4),
          "cold", round(c.get("ms_per_step", 0), 4), "late", round(l.get("ms_per_step", 0), 4), "deep", round(d.get("ms_per_step", 0), 4), "cpu", (o.get("cpu_baseline") or {}).get("value"))
PY
;;
prof)
for c in sand10m fluid20m neo1m snow1m sandblock10m; do bash tools/prof.sh r06 $c > /dev/null 2>&1; grep -E "k_g2p2g|k_vol_p2g|k_grid_update" gpurun_out/prof_r06_$c/kernel_stats.csv | cut -c1-48,120-230; done
;;
tests)
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r06/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r06/pytest_gpu.txt
cp gpurun_out/parity.jsonl gpurun_out/r06/parity.jsonl 2>/dev/null
;;
esac
