#!/bin/bash
mkdir -p gpurun_out/ab
for r in 1 2 3; do
python tools/ab_bench.py early=sparkl_amd/lib/libsparkl_hip.so late=sparkl_amd/lib/ab_pfl.so 2>&1 | tee -a gpurun_out/ab/ab_$1.txt
done
bash tools/pmc.sh late sparkl_amd/lib/ab_pfl.so "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" | tee -a gpurun_out/ab/ab_$1.txt
