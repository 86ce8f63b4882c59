cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -m gpu -x -q > gpurun_out/t_full.txt 2>&1
grep -E "passed|failed|FAILED" gpurun_out/t_full.txt | tail -4
timeout 900 python bench.py > gpurun_out/r03b_bench_sand.json 2> gpurun_out/r03b_bench_sand.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03b_bench_sand_driver.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --clock-wake-ms 0 --no-cpu-baseline > gpurun_out/r03b_bench_sand_driver_nowake.json 2>/dev/null
timeout 600 python bench.py --config neo1m --no-cpu-baseline > gpurun_out/r03b_bench_neo.json 2>/dev/null
timeout 600 python bench.py --config fluid20m --no-cpu-baseline > gpurun_out/r03b_bench_fluid.json 2>/dev/null
bash tools/prof.sh r03b sand10m 32 > /dev/null 2>&1
bash tools/prof.sh r03b fluid20m 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug/run_stats.py 0 640 2000 > gpurun_out/r03b_run_stats.txt 2>&1
python - <<PY
import json
for f in ("r03b_bench_sand","r03b_bench_sand_driver","r03b_bench_sand_driver_nowake","r03b_bench_neo","r03b_bench_fluid"):
    try:
        o=json.load(open("gpurun_out/%s.json"%f))
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f, "ms/step %.4f frac %.3f k %.4f (%.3f)"%(o["ms_per_step"], o.get("step_frac_of_hbm_peak",0), o["roofline"]["avg_launch_ms"], o["roofline"]["frac"]), {a:round(b,4) for a,b in o.get("kernels_avg_launch_ms",{}).items()})
    for k in ("late_collapse","deep_collapse"):
        l=o.get(k)
        if l: print("  ",k, "%.4f frac %.3f"%(l["ms_per_step"], l["step_frac_of_hbm_peak"]), {a:round(b,4) for a,b in l["kernels_avg_launch_ms"].items()}, "kfrac %.3f"%l["roofline"]["frac"], "resorts", l["resorts"], "noop", l.get("noop_launches"), "reorder %.2f"%l["reorder_ms_total"])
    if "cpu_baseline" in o: print("   cpu", o["cpu_baseline"])
PY
