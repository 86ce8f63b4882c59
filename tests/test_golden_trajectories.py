"""Committed golden trajectories (tests/golden/trajectory_*.npz, made by tests/golden/make_golden.py with the f32 oracle and,
as `*_f64`, with the same algorithm evaluated in f64).

CPU: the oracle still reproduces them bit-for-bit (guards the checker against accidental edits) and the scene generators
still produce the same seeded inputs.  GPU: the HIP path matches them under the parity gate of tests/parity.py:
eps(HIP vs f64 golden) <= 1e-4, positions in h, velocities relative to max |v| (the f32 golden columns -- the reference's own
arithmetic -- are printed beside it)."""
import os

import numpy as np
import pytest

from sparkl_amd import scenes
from sparkl_amd.models import (CorotatedLinearElasticity, MaximumStressFailure, NaccPlasticity, NeoHookeanElasticity, RankinePlasticity,
                               SnowPlasticity)

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "sand3_like": (lambda: scenes.sand_column(12, 8, 10, with_block=True), 2.5e-4),
    "neo_hookean": (lambda: scenes.elastic_cube(12), 1.0e-4),
    "fluid": (lambda: scenes.fluid_dam(16, 12, 12), 1.0e-4),
    # heightfield ground and the material models no 3-D example uses (60 substeps so that they act)
    "heightfield": (lambda: _moving(scenes.sand_on_heightfield(10, 6, 8)), 2.5e-4),
    "snow": (lambda: scenes.block_on_floor(10, CorotatedLinearElasticity(1.4e5, 0.2), SnowPlasticity(), density=400.0, v0=(0.0, -8.0, 0.0), dt=2.0e-4, name="snow"), 2.0e-4),
    "rankine": (lambda: scenes.block_on_floor(10, NeoHookeanElasticity(1.0e6, 0.3), RankinePlasticity(1.0e6, 0.3, 5.0e4, 5.0), expand=6.0, v0=(0.0, 0.0, 0.0),
                                              gravity=(0.0, 0.0, 0.0), name="rankine"), 1.0e-4),
    "nacc": (lambda: scenes.block_on_floor(10, NeoHookeanElasticity(2.0e5, 0.3), NaccPlasticity(2.0e5, 0.3, 0.05, True, 1.5, 0.6108652), density=1500.0,
                                           v0=(0.0, -8.0, 0.0), dt=2.0e-4, name="nacc"), 2.0e-4),
    "failure": (lambda: scenes.block_on_floor(10, CorotatedLinearElasticity(1.0e6, 0.3), None, MaximumStressFailure(5.0e4, 4.0e4), expand=6.0, v0=(0.0, 0.0, 0.0),
                                              gravity=(0.0, 0.0, 0.0), name="failure"), 1.0e-4),
}
STEPS = {"sand3_like": (1, 10, 100), "heightfield": (1, 60, 100), "snow": (1, 60), "rankine": (1, 60), "nacc": (1, 60), "failure": (1, 60)}   # default (1, 10)


def _moving(sc):
    sc.particles.velocity["vector"][:, 1] = -3.0
    return sc


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
    for k in STEPS.get(name, (1, 10)):
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
        for k in STEPS.get(name, (1, 10)):
            p.substeps(sc.params, sc.gravity, dt, k - done)
            done = k
            c = p.read_particles()
            p64, v64 = g[f"pos{k}_f64"].astype(np.float64), g[f"vel{k}_f64"].astype(np.float64)
            vmax = np.linalg.norm(v64, axis=1).max()

            def eps(pos, vel):
                return (np.linalg.norm(pos.astype(np.float64) - p64, axis=1).max() / sc.cell_width,
                        np.linalg.norm(vel.astype(np.float64) - v64, axis=1).max() / vmax)
            ex, ev = eps(c.position["point"], c.velocity["vector"])
            fx, fv = eps(g[f"pos{k}"], g[f"vel{k}"])          # the reference's own f32 arithmetic against the same arbiter
            eF = np.abs(c.volume["deformation_gradient"] - g[f"F{k}"]).max()
            print(f"golden {name} k={k}: eps_x {ex:.3e} (f32 oracle {fx:.3e})  eps_v {ev:.3e} (f32 oracle {fv:.3e})  eps_F {eF:.3e}")
            assert ex <= 1e-4 and ev <= 1e-4, (name, k, ex, fx, ev, fv)
            # NACC snaps particles onto the tips / the yield surface by thresholds: a particle sitting on one may take the other
            # branch (the f32 and f64 oracles differ by 6e-3 in F there, test_rare_material_models_parity)
            assert eF <= (1e-2 if name == "nacc" else 1e-3 if k >= 100 else 1e-4), (name, k, eF)
