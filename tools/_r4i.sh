cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_gpu_slab.py -m gpu -x -q -k "adaptive or fluid" > gpurun_out/r4i/pytest.txt 2>&1; tail -6 gpurun_out/r4i/pytest.txt
python tools/ab_bench.py --no-cold-window base=sparkl_amd/lib/ab_v1.so new= base2=sparkl_amd/lib/ab_v1.so new2= > gpurun_out/r4i/ab.txt 2>&1; cat gpurun_out/r4i/ab.txt
python bench.py --config fluid20m --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/r4i/fluid.json 2>/dev/null; python -c "
import json; o=json.loads(open('gpurun_out/r4i/fluid.json').read().strip().splitlines()[-1]); print('fluid', o['ms_per_step'], o['step_frac_of_hbm_peak'], o['kernels_avg_launch_ms'])"
SPARKL_HIP_LIB=$PWD/sparkl_amd/lib/ab_v1.so python bench.py --config fluid20m --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/r4i/fluid_base.json 2>/dev/null; python -c "
import json; o=json.loads(open('gpurun_out/r4i/fluid_base.json').read().strip().splitlines()[-1]); print('fluid base', o['ms_per_step'], o['step_frac_of_hbm_peak'], o['kernels_avg_launch_ms'])"
