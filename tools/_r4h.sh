cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout 1500 python -m pytest tests/test_gpu_full_size_parity.py -m gpu -x -q -k "flowing or fluid" > gpurun_out/r4h/pytest1.txt 2>&1; tail -6 gpurun_out/r4h/pytest1.txt
timeout 1500 python -m pytest tests/test_gpu_slab.py tests/test_golden_trajectories.py tests/test_gpu_parity.py tests/test_checkpoint.py -m gpu -x -q > gpurun_out/r4h/pytest2.txt 2>&1; tail -6 gpurun_out/r4h/pytest2.txt
python tools/probe.py slabs --world 4 --drift 6 > gpurun_out/r4h/soak_slabs.txt 2>&1; tail -8 gpurun_out/r4h/soak_slabs.txt
