#!/bin/bash
# per-launch counters of the first k_g2p2g launches (what is the bump at launches 4-15 made of?)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/early_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 30 --warmup 0 --no-cpu-baseline --no-late-phase --no-kernel-clock > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_g2p2g" not in r["Kernel_Name"]: continue
    per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(per)
names = sorted(next(iter(per.values())))
print("launch " + " ".join(f"{n[3:]:>16s}" for n in names))
for i, k in enumerate(keys[:30]):
    print(f"{i:6d} " + " ".join(f"{per[k].get(n, 0):16.0f}" for n in names))
PY
rm -rf $OUT/trace
