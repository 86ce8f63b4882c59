#!/bin/bash
mkdir -p gpurun_out/fl
python -m pytest tests/test_gpu_parity.py tests/test_golden_trajectories.py tests/test_gpu_full_size_parity.py tests/test_gpu_slab.py -m gpu -x -q -k "fluid or mixed or Fluid or volume or many_models" > gpurun_out/fl/pytest.txt 2>&1
tail -5 gpurun_out/fl/pytest.txt
for r in 1 2 3; do
python tools/ab_bench.py --config fluid20m base=sparkl_amd/lib/ab_base.so new=sparkl_amd/lib/libsparkl_hip.so 2>&1 | tee -a gpurun_out/fl/ab3.txt
done
