#!/bin/bash
# the driver's window (bench.py --steps 20 --warmup 5) on the kernel timeline: what the 20 timed substeps consist of, launch by launch
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/driver_window
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late-phase > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
spk = [i for i, r in enumerate(rows) if "spk::" in r["Kernel_Name"]]
# the timed region: the last 20 k_g2p2g launches and what surrounds them
g = [i for i in spk if "k_g2p2g" in rows[i]["Kernel_Name"]]
first, last = g[-20], g[-1]
# include the grid update before the first fused launch
lo = first - 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
tot = {}
for i in range(lo - 3, min(last + 4, len(rows))):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  +gap %7.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name))
    prev_end = e
    if lo <= i <= last: tot[name] = tot.get(name, 0) + (e - s) / 1e3
span = (int(rows[last]["End_Timestamp"]) - t0) / 1e3
print("GPU span of the 20 substeps: %.1f us = %.4f ms per substep; kernel time in it: %s" % (span, span / 20e3, {k: round(v, 1) for k, v in tot.items()}))
print("bench line:", {k: json.load(open("$OUT/bench.json")).get(k) for k in ("ms_per_step",)})
PY
rm -rf $OUT/trace
