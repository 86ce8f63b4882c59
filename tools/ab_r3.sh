#!/bin/bash
# round-3 A/B on one box: parity suite on the working-tree library, then interleaved bench repeats of the variants given as name=lib.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/ab_r3_parity.txt
cat gpurun_out/ab_r3_parity.txt
for i in 1 2; do timeout 900 python tools/ab_bench.py --steps 100 --warmup 20 "$@" 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/ab_r3_bench.txt
