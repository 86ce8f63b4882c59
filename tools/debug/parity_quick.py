"""eps_v / eps_x of the HIP path against the f64 oracle on three stiff-sand cases (for A/B of arithmetic variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
from tests.parity import particle_errors
cases = [("sand3_like k100", scenes.sand_column(16, 10, 12, with_block=True), 100, 0), ("column12 k40+frame", scenes.sand_column(12, 10, 10), 40, 1), ("sand3_like k10", scenes.sand_column(16, 10, 12, with_block=True), 10, 0)]
for name, sc, nfix, nframe in cases:
    dt = 2.5e-4
    o = Oracle(sc.cell_width, f64=True, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o.load_scene(sc)
    for _ in range(nfix): o.substep_fixed(sc.params, sc.gravity, dt)
    for _ in range(nframe): o.step(sc.params, sc.gravity)
    ref = o.get_particles(); o.close()
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        if nfix: p.substeps(sc.params, sc.gravity, dt, nfix)
        for _ in range(nframe): p.step(sc.params, sc.gravity)
        e = particle_errors(p.read_particles(), ref, sc.cell_width)
    print(f"  {name:20s} eps_v {e['eps_v']:.2e} (mean {e['eps_v_mean']:.2e}) eps_x {e['eps_x']:.2e} eps_F {e['eps_F']:.2e}")
