"""cube_through_sand (configs[4]'s scene) at a few million particles on one pipeline: throughput, re-bins, structure check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.cube_through_sand(240, 60, 120)
print(len(sc.particles), "particles")
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 10)
    p.clock_enable(True, stride=4); p.clock_read(reset=True)
    p.sync(); t0 = time.perf_counter()
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 400)
    p.sync(); ms = 1e3 * (time.perf_counter() - t0) / 400
    k = p.clock_read()
    c = p.grid_check()
    part = p.read_particles()
    v = part.velocity["vector"]
    print(f"{ms:.3f} ms per substep, {len(sc.particles) / ms / 1e6:.2f} G particle-substeps/s, re-bins {int(k.resorts)}, reorders {p.reorder_policy_stats()['reorders']}, C rebuilt {p.step_stats()['c_regathers']}")
    print("grid check:", c)
    print("finite:", bool(np.isfinite(v).all() and np.isfinite(part.position['point']).all()), "failed:", int(part.status["failed"].sum()), "vmax", float(np.abs(v).max()))
