#!/bin/bash
# pmc_steps.sh <tag> <warmup> <steps>: SQ_INSTS_VALU / wave-cycle counters of k_g2p2g for a chosen window of substeps
TAG=$1; W=$2; K=$3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps $K --warmup $W --no-cpu-baseline --no-late-phase --no-kernel-clock > /dev/null 2> $OUT/err
python3 - <<EOF2
import csv, glob, collections
f = glob.glob("$OUT/p/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_g2p2g" in r["Kernel_Name"]]
# per dispatch
by = collections.defaultdict(dict)
for r in rows: by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
print("$TAG dispatches", len(ids))
for i in ids[:: max(1, len(ids) // 12)]:
    c = by[i]; print(i, "VALU %.3e LDS %.3e SALU %.3e wavecyc %.3e wait %.3e" % (c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_WAVE_CYCLES", 0), c.get("SQ_WAIT_ANY", 0)))
EOF2
