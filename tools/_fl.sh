#!/bin/bash
mkdir -p gpurun_out/fl
python -m pytest tests/test_gpu_parity.py tests/test_golden_trajectories.py -m gpu -x -q > gpurun_out/fl/pytest.txt 2>&1
tail -3 gpurun_out/fl/pytest.txt
python tools/probe.py soak --config fluid20m --chunks 5 --chunk 500 2>&1 | tail -8 | cut -c1-200
python tools/ab_bench.py --config fluid20m base=sparkl_amd/lib/ab_base.so new=sparkl_amd/lib/libsparkl_hip.so
