#!/bin/bash
mkdir -p gpurun_out/fl
for r in 1 2 3; do
python tools/ab_bench.py --config fluid20m base=sparkl_amd/lib/libsparkl_hip.so pf130=sparkl_amd/lib/ab_pf130.so pf128=sparkl_amd/lib/ab_pf128.so 2>&1 | tee -a gpurun_out/fl/ab4.txt
done
