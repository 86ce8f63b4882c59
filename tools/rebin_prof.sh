#!/bin/bash
# rocprofv3 kernel stats of the re-bin probe (tools/rebin_probe.py).  Usage: rebin_prof.sh <tag> [sand|fluid]
TAG=${1:-r02}
W=${2:-sand}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rebin_${TAG}_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/rebin_probe.py $W > $OUT/probe.txt 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cat $OUT/probe.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:9.2f}")
PY
