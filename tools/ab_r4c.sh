#!/bin/bash
# round 4: instruction diet of k_g2p2g (FMA chains, det F - 1 from the return map, descriptor rows as pairs, 32-bit block offsets, dead-lane init)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
for rep in 1 2; do
python tools/ab_bench.py v0=sparkl_amd/lib/ab_v0.so diet= nomerge=sparkl_amd/lib/ab_nomerge.so >> gpurun_out/r4c/ab.txt 2>&1
done
cat gpurun_out/r4c/ab.txt
bash tools/pmc_valu.sh v0 sparkl_amd/lib/ab_v0.so > gpurun_out/r4c/pmc.txt 2>&1
bash tools/pmc_valu.sh diet "" >> gpurun_out/r4c/pmc.txt 2>&1
bash tools/pmc_valu.sh nomerge sparkl_amd/lib/ab_nomerge.so >> gpurun_out/r4c/pmc.txt 2>&1
cat gpurun_out/r4c/pmc.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4c/pytest.txt 2>&1; tail -3 gpurun_out/r4c/pytest.txt
