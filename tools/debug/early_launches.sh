#!/bin/bash
# durations of the first k_g2p2g launches after pipeline creation (clock ramp / first touch?)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/early
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 60 --warmup 0 --no-cpu-baseline --no-late-phase --no-kernel-clock > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_g2p2g" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print(" ".join(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.0f}" for r in rows[:70]))
print("start offsets ms:", " ".join(f"{(int(r['Start_Timestamp'])-t0)/1e6:.1f}" for r in rows[:70:5]))
PY
rm -rf $OUT/trace
