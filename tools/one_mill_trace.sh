#!/bin/bash
# one_mill_trace.sh: rocprofv3 --kernel-trace of the two 1 M configurations (neo1m, snow1m) over 400 substeps through tools/slab_trace.py: where a substep of a
# short-launch scene goes (kernel, grid update, gaps, re-bins and block reorders).  GPU box: gpurun -- bash tools/one_mill_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
for c in neo1m snow1m; do
rm -rf /tmp/tr_$c
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -- python3 $R/bench.py --config $c --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-clock --no-cold-window > /tmp/tr_$c.json 2>/dev/null
tail -1 /tmp/tr_$c.json | cut -c1-260
python3 $R/tools/slab_trace.py /tmp/tr_$c --tail-fraction 0.3 | head -16
done
