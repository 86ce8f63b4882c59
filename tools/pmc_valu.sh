#!/bin/bash
# pmc_valu.sh <tag> <lib-or-empty> [ENV=VAL ...]: one SQ counter pass of bench.py (8 substeps) for one library variant: VALU / LDS instruction
# counts and busy cycles of the particle kernels per launch (GPU box, via gpurun)
TAG=$1; LIB=$2; shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
( [ -n "$LIB" ] && export SPARKL_HIP_LIB=$R/$LIB
for kv in "$@"; do export $kv; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-late-phase --no-kernel-clock --no-cold-window --clock-wake-ms 0 > /dev/null 2> $OUT/pmc_a.err )
python3 $R/tools/tools_pmc_summary.py $OUT | grep g2p2g > $OUT/summary.txt
echo "== $TAG"; cat $OUT/summary.txt
