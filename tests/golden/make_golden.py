"""Generates tests/golden/trajectory_*.npz with the f32 oracle (oracle/mpm_oracle.hpp).

The reference holds no dynamics fixture and cannot be run here (Rust, F2/F4), so these vectors are produced by
the restated CPU path itself: inputs are rebuilt from the seeded scene generators, outputs are the particle state after
1, 10 (60, 100) fixed substeps -- of the f32 oracle (the reference's arithmetic) and, as `*_f64`, of the same algorithm
evaluated in f64 (the arbiter of the parity gate, tests/parity.py).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import Oracle  # noqa: E402
from sparkl_amd import scenes  # noqa: E402
from sparkl_amd.models import (CorotatedLinearElasticity, MaximumStressFailure, NaccPlasticity, NeoHookeanElasticity, RankinePlasticity,  # noqa: E402
                               SnowPlasticity)

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


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = sys.argv[1:]
    for name, (build, dt) in CASES.items():
        if only and name not in only:
            continue
        sc = build()
        o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes)
        o.load_scene(sc)
        o64 = Oracle(sc.cell_width, f64=True, domain_log2_bytes=sc.domain_log2_bytes)
        o64.load_scene(sc)
        out = {"dt": dt, "n": len(sc.particles), "pos0": sc.particles.position["point"]}
        done = 0
        for k in STEPS.get(name, (1, 10)):
            for _ in range(k - done):
                o.substep_fixed(sc.params, sc.gravity, dt)
                o64.substep_fixed(sc.params, sc.gravity, dt)
            done = k
            c = o.get_particles()
            c64 = o64.get_particles()
            out[f"pos{k}_f64"] = c64.position["point"]
            out[f"vel{k}_f64"] = c64.velocity["vector"]
            out[f"pos{k}"] = c.position["point"]
            out[f"vel{k}"] = c.velocity["vector"]
            out[f"F{k}"] = c.volume["deformation_gradient"]
            out[f"C{k}"] = c.velocity_gradient["m"]
            out[f"data{k}"] = np.stack([c.data["plastic_hardening"], c.data["log_vol_gain"], c.volume["plastic_deformation_gradient_det"],
                                        c.data["elastic_hardening"], c.data["nacc_alpha"], c.phase["phase"]], 1)
        np.savez_compressed(os.path.join(out_dir, f"trajectory_{name}.npz"), **out)
        print(name, len(sc.particles), "particles")


if __name__ == "__main__":
    main()
