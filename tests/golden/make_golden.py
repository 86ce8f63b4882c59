"""Generates tests/golden/trajectory_*.npz with the f32 oracle (oracle/mpm_oracle.hpp).

The reference holds no dynamics fixture and cannot be run here (Rust, F2/F4), so these vectors are produced by
the restated CPU path itself: inputs are rebuilt from the seeded scene generators, outputs are the particle state after
1 and 10 fixed substeps.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import Oracle  # noqa: E402
from sparkl_amd import scenes  # noqa: E402

CASES = {
    "sand3_like": (lambda: scenes.sand_column(12, 8, 10, with_block=True), 2.5e-4),
    "neo_hookean": (lambda: scenes.elastic_cube(12), 1.0e-4),
    "fluid": (lambda: scenes.fluid_dam(16, 12, 12), 1.0e-4),
}


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (build, dt) in CASES.items():
        sc = build()
        o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes)
        o.load_scene(sc)
        out = {"dt": dt, "n": len(sc.particles), "pos0": sc.particles.position["point"]}
        done = 0
        for k in (1, 10):
            for _ in range(k - done):
                o.substep_fixed(sc.params, sc.gravity, dt)
            done = k
            c = o.get_particles()
            out[f"pos{k}"] = c.position["point"]
            out[f"vel{k}"] = c.velocity["vector"]
            out[f"F{k}"] = c.volume["deformation_gradient"]
            out[f"C{k}"] = c.velocity_gradient["m"]
            out[f"data{k}"] = np.stack([c.data["plastic_hardening"], c.data["log_vol_gain"], c.volume["plastic_deformation_gradient_det"]], 1)
        np.savez_compressed(os.path.join(out_dir, f"trajectory_{name}.npz"), **out)
        print(name, len(sc.particles), "particles")


if __name__ == "__main__":
    main()
