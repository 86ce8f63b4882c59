"""Re-bin cost probe: a 10M sand column and a 20M fluid dam given a uniform drift so that re-bins recur; prints per-re-bin ms
(SparklKernelClock.sort_ms / resorts, steady state: the first re-bin, which allocates, is excluded)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
which = sys.argv[1] if len(sys.argv) > 1 else "sand"
if which == "sand":
    sc = scenes.sand_column(200, 250, 200); sc.particles.velocity["vector"][:, 0] = 8.0; dt = sc.fixed_dt; n = 300
else:
    sc = scenes.fluid_dam(344, 240, 244); sc.particles.velocity["vector"][:, 0] = 12.0; dt = 5.0e-4; n = 160
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, dt, 4)
    p.clock_enable(True, stride=1)
    p.clock_read(reset=True)
    p.sync(); t0 = time.perf_counter()
    p.substeps(sc.params, sc.gravity, dt, n)
    p.sync(); ms = 1e3 * (time.perf_counter() - t0)
    k = p.clock_read()
    print(f"{which}: {len(sc.particles)} particles, {n} substeps {ms / n:.3f} ms/substep, re-bins {int(k.resorts)} total {k.sort_ms:.2f} ms -> {k.sort_ms / max(int(k.resorts), 1):.3f} ms each; blocks {p.num_particle_blocks()} / {p.num_active_blocks()}")
