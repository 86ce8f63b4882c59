#!/bin/bash
# pmc.sh <tag> <lib-or-""> "<counters>" [bench.py flags ...]: one rocprofv3 counter pass of bench.py for one library variant (GPU box, via
# gpurun); per-launch means of the particle / node kernels -> gpurun_out/pmc_<tag>/summary.txt.  Counters default to the issue / LDS set;
# HBM traffic needs its own passes ("FETCH_SIZE", then "WRITE_SIZE": tools/prof.sh does both and applies nothing -- the gfx950 correction
# FETCH x 2 is applied where the numbers are used, profiles/pmc_traffic.json).  Environment variables pass through (SPARKL_HIP_MSET=0 ...).
TAG=$1; LIB=$2; CNT=${3:-"SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS"}
shift $(( $# < 3 ? $# : 3 ))   # (a plain `shift 3` with fewer arguments shifts NOTHING and the tag would reach bench.py as a flag)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
( [ -n "$LIB" ] && export SPARKL_HIP_LIB=$R/$LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-late-phase --no-kernel-clock --no-cold-window --no-strong-projection --clock-wake-ms 0 "$@" > /dev/null 2> $OUT/pmc_a.err ) || { echo "pmc.sh $TAG: rocprofv3 / bench.py FAILED (rc $?)"; tail -5 $OUT/pmc_a.err; exit 1; }
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
rm -rf $OUT/pmc_a
echo "== $TAG"; cat $OUT/summary.txt
