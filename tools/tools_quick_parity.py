"""Quick GPU-vs-oracle check used while optimising kernels (not part of the test suite)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
from tests.parity import particle_errors
which = sys.argv[1:] or ["sand", "elastic", "fluid"]
T = max(1, min(16, len(os.sched_getaffinity(0))))
for name in which:
    sc = {"sand": lambda: scenes.sand_column(24, 12, 16, with_block=True), "elastic": lambda: scenes.elastic_cube(20),
          "fluid": lambda: scenes.fluid_dam(24, 16, 16)}[name]()
    o = Oracle(sc.cell_width, threads=T); o.load_scene(sc)
    o64 = Oracle(sc.cell_width, f64=True, threads=T); o64.load_scene(sc)
    n = 40
    t = time.time()
    for _ in range(n):
        o.substep_fixed(sc.params, sc.gravity, sc.fixed_dt); o64.substep_fixed(sc.params, sc.gravity, sc.fixed_dt)
    with MpmPipeline(sc.cell_width) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, n)
        got = p.read_particles()
    e = particle_errors(got, o.get_particles(), sc.cell_width); f = particle_errors(o.get_particles(), o64.get_particles(), sc.cell_width)
    e64 = particle_errors(got, o64.get_particles(), sc.cell_width)
    print(f"{name:8s} n={len(sc.particles)} k={n}: vs f32 eps_x {e['eps_x']:.2e} eps_v {e['eps_v']:.2e} eps_F {e['eps_F']:.2e} | vs f64 eps_v {e64['eps_v']:.2e} | floor eps_x {f['eps_x']:.2e} eps_v {f['eps_v']:.2e} | failed_mismatch {e['failed_mismatch']} ({time.time()-t:.1f}s)")
