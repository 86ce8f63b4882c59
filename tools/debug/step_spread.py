"""Isolate the step() divergence on a fast expanding cube: adaptive step vs the oracle, and the oracle's dt sequence replayed
through fixed substeps."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
from tests.parity import particle_errors

mult = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
def build():
    sc = scenes.elastic_cube(10)
    sc.particles.velocity["vector"] *= mult
    sc.params.dt = 4.0e-3
    return sc
sc = build()
o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o.load_scene(sc)
o2 = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o2.load_scene(sc)
pa = MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes); pa.load_scene(sc)
pb = MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes); pb.load_scene(sc)
for frame in range(4):
    o.step(sc.params, sc.gravity)
    dts = list(o.last_dts())
    t, tim = pa.step(sc.params, sc.gravity, timings=True)
    for d in dts:
        pb.substeps(sc.params, sc.gravity, float(d), 1)
        o2.substep_fixed(sc.params, sc.gravity, float(d))
    ea = particle_errors(pa.read_particles(), o.get_particles(), sc.cell_width)
    eb = particle_errors(pb.read_particles(), o.get_particles(), sc.cell_width)
    ec = particle_errors(o2.get_particles(), o.get_particles(), sc.cell_width)
    print(f"frame {frame}: oracle dts {['%.6e' % d for d in dts]} hip substeps {tim.num_substeps} resorts {tim.num_resorts}")
    print(f"   step() vs oracle   eps_x {ea['eps_x']:.3e} eps_v {ea['eps_v']:.3e}")
    print(f"   replayed dts (HIP) eps_x {eb['eps_x']:.3e} eps_v {eb['eps_v']:.3e}")
    print(f"   replayed dts (oracle fixed) eps_x {ec['eps_x']:.3e} eps_v {ec['eps_v']:.3e}")
