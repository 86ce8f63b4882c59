"""Committed golden trajectories (tests/golden/trajectory_*.npz, made by tests/golden/make_golden.py with the f32 oracle).

CPU: the oracle still reproduces them bit-for-bit (guards the checker against accidental edits) and the scene generators
still produce the same seeded inputs.  GPU: the HIP path matches them within the parity tolerance."""
import os

import numpy as np
import pytest

from sparkl_amd import scenes

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "sand3_like": (lambda: scenes.sand_column(12, 8, 10, with_block=True), 2.5e-4),
    "neo_hookean": (lambda: scenes.elastic_cube(12), 1.0e-4),
    "fluid": (lambda: scenes.fluid_dam(16, 12, 12), 1.0e-4),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    from oracle import Oracle
    g = np.load(os.path.join(GOLD, f"trajectory_{name}.npz"))
    build, dt = CASES[name]
    sc = build()
    np.testing.assert_array_equal(sc.particles.position["point"], g["pos0"])   # seeded generator (PCG32) is stable
    o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes)
    o.load_scene(sc)
    done = 0
    for k in (1, 10):
        for _ in range(k - done):
            o.substep_fixed(sc.params, sc.gravity, dt)
        done = k
        c = o.get_particles()
        np.testing.assert_array_equal(c.position["point"], g[f"pos{k}"])
        np.testing.assert_array_equal(c.velocity["vector"], g[f"vel{k}"])
        np.testing.assert_array_equal(c.volume["deformation_gradient"], g[f"F{k}"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_golden(name):
    from sparkl_amd.pipeline import MpmPipeline
    g = np.load(os.path.join(GOLD, f"trajectory_{name}.npz"))
    build, dt = CASES[name]
    sc = build()
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        done = 0
        for k in (1, 10):
            p.substeps(sc.params, sc.gravity, dt, k - done)
            done = k
            c = p.read_particles()
            vmax = np.linalg.norm(g[f"vel{k}"], axis=1).max()
            ex = np.linalg.norm(c.position["point"] - g[f"pos{k}"], axis=1).max() / sc.cell_width
            ev = np.linalg.norm(c.velocity["vector"] - g[f"vel{k}"], axis=1).max() / vmax
            eF = np.abs(c.volume["deformation_gradient"] - g[f"F{k}"]).max()
            # 1e-4 on positions; velocities of the stiff sand sit on its f32 noise floor
            assert ex <= 1e-4 and eF <= 1e-4, (name, k, ex, ev, eF)
            assert ev <= (1e-3 if name == "sand3_like" else 1e-4), (name, k, ev)
