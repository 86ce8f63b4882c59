import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 90
sc = scenes.elastic_cube(16)
sc.particles.velocity["vector"] *= 40.0
dt = 2.0e-4
got = {}
for mode in ("0", "1"):
    os.environ["SPARKL_HIP_DEFER_C"] = mode
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.clock_enable(True)
        p.substeps(sc.params, sc.gravity, dt, nsub)
        got[mode] = (p.read_particles(), p.step_stats(), int(p.clock_read().resorts))
a, b = got["0"][0], got["1"][0]
Ca = a.velocity_gradient.view(np.float32).reshape(len(a.velocity_gradient), -1)
Cb = b.velocity_gradient.view(np.float32).reshape(len(b.velocity_gradient), -1)
d = np.abs(Ca - Cb)
print("stats", got["1"][1], "resorts", got["1"][2], got["0"][2])
print("max |dC| per component:", d.max(axis=0), "cmax", np.abs(Ca).max())
bad = np.argsort(-d.max(axis=1))[:5]
for i in bad:
    print(i, "C0", Ca[i], "\n   C1", Cb[i], "\n   x", a.position[i], b.position[i], "v", a.velocity[i], b.velocity[i], "status", a.status[i])
print("particles with |dC| > 1e-3 cmax:", int((d.max(axis=1) > 1e-3 * np.abs(Ca).max()).sum()), "of", len(Ca))
