#!/bin/bash
# round 4: which parts of the diet pay: descriptor rows as pairs (default) vs nine float rows (rows9), merge of split halves off (nomerge)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
for rep in 1 2; do
python tools/ab_bench.py --no-cold-window v0=sparkl_amd/lib/ab_v0.so diet= rows9=sparkl_amd/lib/ab_rows9.so rows9nm=sparkl_amd/lib/ab_rows9nm.so nomerge=sparkl_amd/lib/ab_nomerge.so >> gpurun_out/r4d/ab.txt 2>&1
done
cat gpurun_out/r4d/ab.txt
timeout 600 python -m pytest tests/test_gpu_slab.py -m gpu -x -q -k "comm_init_all or aborts or world_of_one" > gpurun_out/r4d/pytest.txt 2>&1; tail -5 gpurun_out/r4d/pytest.txt
SPARKL_HIP_LIB=$PWD/sparkl_amd/lib/ab_rows9nm.so SPARKL_HIP_LIB_AB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sand or sheared or rebin" > gpurun_out/r4d/pytest_nm.txt 2>&1; tail -3 gpurun_out/r4d/pytest_nm.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late-phase > gpurun_out/r4d/bench_driver.json 2>gpurun_out/r4d/bench_driver.err; python -c "
import json; o=json.loads(open('gpurun_out/r4d/bench_driver.json').read().strip().splitlines()[-1]); print('driver', o['ms_per_step'], o['step_frac_of_hbm_peak'], 'cold', o.get('cold_window'))"
