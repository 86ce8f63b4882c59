#!/bin/bash
# counters of k_g2p2g deep in the flowing phase (substeps 2000..2400 of the 10M column, tools/debug/late_probe.py)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/late_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/debug/late_probe.py > /dev/null 2> $OUT/err_$name.txt
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_g2p2g" in r["Kernel_Name"]]
    # launches behind an abort flag do nothing: drop them (discriminator: the pass's instruction / busy counter against its median)
    disc = {int(r["Dispatch_Id"]): float(r["Counter_Value"]) for r in rows if r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES")}
    med = sorted(disc.values())[len(disc) // 2]
    rows = [r for r in rows if disc[int(r["Dispatch_Id"])] > 0.3 * med]
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))
    n = len(ids)
    early, late = set(ids[20:120]), set(ids[-200:])
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in early: tot["early (launches 20..120)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if d in late: tot["late (last 200)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(k, {c: "%.3e" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
rm -rf $OUT/pmc_*
