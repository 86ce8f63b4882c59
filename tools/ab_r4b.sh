#!/bin/bash
# round 4: the pair return map (MSET=2) against the model-specialised kernel (MSET=1), three prefetch positions; parity of the pair path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
SPARKL_HIP_MSET=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sand or sheared or column or rebin or deferred" > gpurun_out/r4b/pytest_pair.txt 2>&1; tail -5 gpurun_out/r4b/pytest_pair.txt
for rep in 1 2; do
python tools/ab_bench.py mset1=,SPARKL_HIP_MSET=1 pair=,SPARKL_HIP_MSET=2 pair_pf3=sparkl_amd/lib/ab_pf3.so,SPARKL_HIP_MSET=2 pair_pf1=sparkl_amd/lib/ab_pf1.so,SPARKL_HIP_MSET=2 >> gpurun_out/r4b/ab.txt 2>&1
done
cat gpurun_out/r4b/ab.txt
