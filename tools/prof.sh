#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box through gpurun).  Usage: prof.sh <tag> [config] [steps]
# Every profiled command runs under `timeout`: a stuck rocprofv3 must not eat the GPU budget.
TAG=${1:-r03}
CFG=${2:-sand10m}
STEPS=${3:-32}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${TAG}_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --no-cpu-baseline --no-late-phase --no-kernel-clock --no-cold-window --no-strong-projection"   # one window: the averages are the headline window's
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps $STEPS --warmup 8 $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --steps 8 --warmup 8 $ARGS > /dev/null 2> $OUT/pmc_$name.err
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# keep the merged-back payload small: the raw traces are not needed
rm -rf $OUT/trace $OUT/pmc_*/
ls -la $OUT
