#!/bin/bash
# idle time between consecutive kernels of the fused batch (end of one -> start of the next), from a rocprofv3 kernel trace
CFG=${1:-neo1m}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gaps_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 64 --warmup 16 --no-cpu-baseline --no-late-phase --no-kernel-clock --clock-wake-ms 0 > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob, numpy as np
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "spk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 100 substeps' worth of launches
rows = rows[-260:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = np.array(gaps)
print("$CFG: kernels", len(rows), "median duration us", np.median(durs), "| gap us: median %.2f mean %.2f p90 %.2f max %.1f; busy fraction %.3f" % (np.median(g), g.mean(), np.quantile(g, 0.9), g.max(), sum(durs) / (sum(durs) + g[g > 0].sum())))
PY
rm -rf $OUT/trace
