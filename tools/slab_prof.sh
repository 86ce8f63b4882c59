#!/bin/bash
# slab_prof.sh <tag> [probe flags...]: rocprofv3 --kernel-trace of tools/probe.py slabs (GPU box, via gpurun) -> gpurun_out/<tag>_trace.txt (tools/slab_trace.py)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/probe.py slabs "$@" > $OUT/probe.txt 2> $OUT/probe.err
python3 $R/tools/slab_trace.py $OUT/trace > $R/gpurun_out/${TAG}_trace.txt 2>&1
grep -E "one pipeline|substeps" $OUT/probe.txt | cut -c1-200 >> $R/gpurun_out/${TAG}_trace.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cat $R/gpurun_out/${TAG}_trace.txt
