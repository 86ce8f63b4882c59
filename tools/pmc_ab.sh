#!/bin/bash
# pmc_ab.sh <tag> <lib-or-empty> [ENV=VAL ...]: two SQ counter passes of bench.py for one library variant (GPU box, via gpurun)
TAG=$1; LIB=$2; shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
[ -n "$LIB" ] && export SPARKL_HIP_LIB=$R/$LIB
for kv in "$@"; do export $kv; done
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-late-phase --no-kernel-clock > /dev/null 2> $OUT/pmc_$name.err
done
python3 $R/tools/tools_pmc_summary.py $OUT | grep g2p2g > $OUT/summary.txt
echo "== $TAG"; cat $OUT/summary.txt
