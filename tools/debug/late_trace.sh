#!/bin/bash
# per-launch durations of the kernels deep in the flowing phase (substeps 2000..2400 of the 10M column)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/late_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/debug/late_probe.py > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import csv, glob, numpy as np, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "spk::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 400 substeps: take launches after the 2000th k_g2p2g-with-work
g = [i for i, r in enumerate(rows) if "k_g2p2g" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
start = g[-390]
sel = rows[start:]
by = collections.defaultdict(list)
for r in sel:
    name = r["Kernel_Name"].split("(")[0].replace("void spk::", "").replace("spk::", "")[:40]
    by[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
print("window %.1f ms, kernels busy %.1f ms" % ((t1 - t0) / 1e6, sum(sum(v) for v in by.values()) / 1e3))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print(f"{k:42s} n {len(v):5d} total {v.sum() / 1e3:8.2f} ms  mean {v.mean():8.1f} us  p10 {np.quantile(v, 0.1):7.1f} p50 {np.median(v):7.1f} p90 {np.quantile(v, 0.9):7.1f}  (no-op launches < 20 us: {(v < 20).sum()})")
PY
rm -rf $OUT/trace
