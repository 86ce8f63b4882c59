import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
from tests.parity import particle_errors
sc = scenes.fluid_dam(24, 16, 16)
for dt, n in ((1.0e-4, 30), (2.0e-3, 6), (6.0e-3, 3)):
    o = Oracle(sc.cell_width, f64=True, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o.load_scene(sc)
    for _ in range(n): o.substep_fixed(sc.params, sc.gravity, dt)
    ref = o.get_particles(); o.close()
    for mode in ("fused", "unfused"):
        if mode == "unfused": os.environ["SPARKL_HIP_NO_FUSE"] = "1"
        else: os.environ.pop("SPARKL_HIP_NO_FUSE", None)
        with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
            p.load_scene(sc); p.clock_enable(True)
            p.substeps(sc.params, sc.gravity, dt, n)
            e = particle_errors(p.read_particles(), ref, sc.cell_width)
            print(f"dt {dt:g} x {n} {mode:8s}: eps_x {e['eps_x']:.2e} eps_v {e['eps_v']:.2e} eps_F {e['eps_F']:.2e} resorts {int(p.clock_read().resorts)}")
