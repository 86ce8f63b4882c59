#!/bin/bash
# sensitivity + counters: bench A/B of probe variants, then SQ counter passes of two variants
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 900 python tools/ab_bench.py --steps 100 --warmup 20 --no-late-phase new= dv200=sparkl_amd/lib/ab_dv200.so dl16=sparkl_amd/lib/ab_dl16.so scat=sparkl_amd/lib/ab_scat.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/ab_r3b_bench.txt
bash tools/pmc_ab.sh r3_scat sparkl_amd/lib/ab_scat.so 2>&1 | tail -3
bash tools/pmc_ab.sh r3_new "" 2>&1 | tail -3
