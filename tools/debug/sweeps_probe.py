import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
f = np.float32
def jac(pp, qq, pq, rp, rq, skip):
    if abs(pq) > skip:
        zeta = f((qq - pp) * f(f(1) / f(f(2) * pq)))
        t = f(np.copysign(f(1), zeta) * f(f(1) / f(abs(zeta) + np.sqrt(f(f(1) + zeta * zeta)))))
        c = f(f(1) / np.sqrt(f(f(1) + t * t))); s = f(c * t)
        pp = f(pp - t * pq); qq = f(qq + t * pq); pq = f(0)
        nrp = f(c * rp - s * rq); rq = f(s * rp + c * rq); rp = nrp
    return pp, qq, pq, rp, rq
def sweeps(S, TOL_ABS=1e-6, FLOOR=1e-10):
    s00, s11, s22, s01, s02, s12 = S[0, 0], S[1, 1], S[2, 2], S[0, 1], S[0, 2], S[1, 2]
    scale = max(abs(x) for x in (s00, s11, s22, s01, s02, s12))
    tol = max(min(f(1e-2) * scale, max(f(TOL_ABS), f(1e-5) * scale)), f(FLOOR)); skip = f(1e-3) * tol
    more = max(abs(s01), abs(s02), abs(s12)) > tol
    n = 0
    while n < 8 and more:
        s00, s11, s01, s02, s12 = jac(s00, s11, s01, s02, s12, skip)
        s00, s22, s02, s01, s12 = jac(s00, s22, s02, s01, s12, skip)
        s11, s22, s12, s01, s02 = jac(s11, s22, s12, s01, s02, skip)
        more = max(abs(s01), abs(s02)) > tol
        n += 1
    return n, scale
sc = scenes.sand_column(40, 40, 40)
for nsub in (12, 100, 300):
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, nsub)
        F = p.read_particles().volume["deformation_gradient"].reshape(-1, 3, 3).transpose(0, 2, 1).astype(f)   # column-major -> [row][col]
        ids = p.read_sorted_particle_ids()
    F = F[ids][:64 * 200]
    E = F - np.eye(3, dtype=f)
    for TOL_ABS in (1e-6, 3e-6, 1e-5):
        ns, scs = [], []
        for e in E:
            S = (e + e.T + e.T @ e).astype(f)
            n, scv = sweeps(S, TOL_ABS); ns.append(n); scs.append(scv)
        ns = np.array(ns); scs = np.array(scs)
        wave_max = ns.reshape(-1, 64).max(axis=1)
        print(f"after {nsub} substeps, tol_abs {TOL_ABS:g}: strain scale median {np.median(scs):.2e}; per particle {np.bincount(ns)}; per wave (max of 64) {np.bincount(wave_max)} mean {wave_max.mean():.2f}")
    print("  sample S/scale:", np.round((E[5] + E[5].T) / scs[5], 3).tolist())
