import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
from tests.parity import particle_errors
sc = scenes.fluid_dam(24, 16, 16)
sc.params.dt = 2.0e-2; sc.params.max_substep_dt = 6.0e-3
nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
o = Oracle(sc.cell_width, f64=True, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o.load_scene(sc)
refs = []
for _ in range(nframes):
    o.step(sc.params, sc.gravity); refs.append((o.get_particles(), list(o.last_dts())))
o.close()
for mode in ("1", "0"):
    os.environ["SPARKL_HIP_FLUID_STEP_FUSE"] = mode
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        for f in range(nframes):
            t, tim = p.step(sc.params, sc.gravity, timings=True)
            e = particle_errors(p.read_particles(), refs[f][0], sc.cell_width)
            print(f"fuse={mode} frame {f}: substeps {tim.num_substeps} (oracle {len(refs[f][1])}: {[round(x, 5) for x in refs[f][1]]}) eps_x {e['eps_x']:.2e} eps_v {e['eps_v']:.2e} eps_F {e['eps_F']:.2e} stats {p.step_stats()['fluid_fused_substeps']}/{p.step_stats()['fluid_fused_mismatches']} resorts {tim.num_resorts}")
