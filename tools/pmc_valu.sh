#!/bin/bash
# pmc_valu.sh <lib> ...: SQ_INSTS_VALU / SQ_INSTS_SALU / VMEM per launch of k_g2p2g for each library variant (one counter pass each)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  OUT=$R/gpurun_out/pmcv_$(basename $LIB .so)
  rm -rf $OUT; mkdir -p $OUT
  export SPARKL_HIP_LIB=$R/$LIB
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_a -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-late-phase --no-kernel-clock --clock-wake-ms 0 > /dev/null 2> $OUT/err.txt
  echo "== $LIB"; python3 $R/tools/tools_pmc_summary.py $OUT | grep g2p2g
  rm -rf $OUT/pmc_a
done
