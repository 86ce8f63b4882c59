#!/bin/bash
# round 4, first A/B on one box: model-specialised k_g2p2g (MSET) and the occupancy variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
hipcc --offload-arch=gfx950 -O3 -o /tmp/pl tools/microbench/permlane_swap.hip 2>/dev/null && /tmp/pl > gpurun_out/r4a/permlane.txt 2>&1
for rep in 1 2; do
python tools/ab_bench.py base=,SPARKL_HIP_MSET=0 mset1= mb3=sparkl_amd/lib/ab_mb3.so mb2=sparkl_amd/lib/ab_mb2.so >> gpurun_out/r4a/ab.txt 2>&1
done
cat gpurun_out/r4a/permlane.txt gpurun_out/r4a/ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.txt 2>&1; tail -5 gpurun_out/r4a/pytest.txt
