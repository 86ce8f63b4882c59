import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from sparkl_amd.models import ParticleModel, CorotatedLinearElasticity, DruckerPragerPlasticity
from oracle import Oracle
def scene(n_models, mix):
    sc = scenes.sand_column(12, 10, 10)
    for k in range(1, n_models):
        E = 1.0e7 * (1.0 + 0.05 * k)
        m = ParticleModel.with_plasticity(CorotatedLinearElasticity(E, 0.2), DruckerPragerPlasticity(E, 0.2)) if (k % 2 or not mix) else ParticleModel.new(CorotatedLinearElasticity(E, 0.2))
        sc.models.insert(m)
    sc.particles.status["model_index"] = (np.arange(len(sc.particles)) % n_models).astype(sc.particles.status["model_index"].dtype)
    return sc
for n_models, mix, nfix, nframe in [(1, False, 40, 0), (4, False, 40, 0), (4, True, 40, 0), (4, True, 0, 1), (1, False, 0, 1), (1, False, 40, 1)]:
    sc = scene(n_models, mix)
    dt = 2.5e-4
    os_ = []
    for f64 in (False, True):
        o = Oracle(sc.cell_width, f64=f64, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o.load_scene(sc); os_.append(o)
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        for o in os_:
            for _ in range(nfix): o.substep_fixed(sc.params, sc.gravity, dt)
            for _ in range(nframe): o.step(sc.params, sc.gravity)
        if nfix: p.substeps(sc.params, sc.gravity, dt, nfix)
        for _ in range(nframe): p.step(sc.params, sc.gravity)
        g = p.read_particles()
    r32, r64 = os_[0].get_particles(), os_[1].get_particles()
    vmax = np.abs(r64.velocity["vector"]).max()
    def ev(a, b): 
        d = a.velocity["vector"].astype(np.float64) - b.velocity["vector"].astype(np.float64)
        return np.linalg.norm(d, axis=1).max() / vmax, d.mean(axis=0) / vmax
    e1, m1 = ev(g, r64); e2, m2 = ev(r32, r64)
    print(f"models {n_models} mix {mix} fixed {nfix} frames {nframe}: vmax {vmax:.4f} hip-vs-f64 {e1:.2e} mean-signed {m1}  |  f32-vs-f64 {e2:.2e} mean-signed {m2}")
    for o in os_: o.close()
