#!/bin/bash
# r06_soak.sh: repeat the multi-process exchange tests (flakiness) and run the snow view for 6 000 substeps with the structure validator -> gpurun_out/r06/soak.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_slab_procs.py -q -m gpu 2>&1 | grep -E "passed|failed"; done
python - <<'PY'
import time, numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from sparkl_amd.models import CorotatedLinearElasticity, SnowPlasticity
sc = scenes.block_on_floor(100, CorotatedLinearElasticity(1.4e5, 0.2), SnowPlasticity(), density=400.0, v0=(0.5, -6.0, 0.2), dt=2.0e-4, name="snow")
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    t0 = time.time()
    for k in range(12):
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 500)
        chk = p.grid_check()
        g = p.read_particles()
        jp = g.volume["plastic_deformation_gradient_det"]
        bad = sum(v for name, v in chk.items() if name not in ("node_blocks", "particle_blocks"))
        nb, npb = chk["node_blocks"], chk["particle_blocks"]
        print(f"snow view, {len(sc.particles)} particles, {500 * (k + 1)} substeps: validator {bad} violations ({nb} node blocks, {npb} particle blocks), finite {bool(np.isfinite(g.position['point']).all())}, "
              f"Jp min {jp.min():.4f}, compacted {(jp < 0.999).mean():.3f}, c_regathers {p.step_stats()['c_regathers']}, {1e3 * (time.time() - t0) / (500 * (k + 1)):.4f} ms per substep incl. read-backs")
PY
} > gpurun_out/r06/soak.txt 2>&1
cat gpurun_out/r06/soak.txt
