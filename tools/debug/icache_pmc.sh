R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-late-phase --no-kernel-clock > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_g2p2g" in r["Kernel_Name"] or "k_grid_update" in r["Kernel_Name"]:
            tot[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(tot.items()): print(k, "%.4g" % (sum(v) / len(v)), len(v))
PY
tail -3 $OUT/err.txt
