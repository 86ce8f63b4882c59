"""Grid-update cost at 10M particles with the sand3 ground (41 x 41 heightfield under the whole column) and with a triangle mesh,
next to the cuboid floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes, abi
from sparkl_amd.pipeline import MpmPipeline
f32 = np.float32
def run(name, sc):
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 8)
        p.clock_enable(True, stride=4); p.clock_read(reset=True)
        p.sync(); t0 = time.perf_counter()
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 120)
        p.sync(); ms = 1e3 * (time.perf_counter() - t0) / 120
        k = p.clock_read()
        print(f"{name:12s}: {ms:.3f} ms per substep, grid update {k.grid_update_ms / max(k.grid_update_launches, 1):.4f} ms per launch, g2p2g {k.g2p2g_ms / max(k.g2p2g_launches, 1):.4f}", flush=True)
base = scenes.sand_column(200, 250, 200)
run("cuboid floor", base)
sc = scenes.sand_column(200, 250, 200)
n = 40
i = np.arange(n + 1, dtype=f32)
heights = np.repeat((-np.sin(i * f32(np.pi) / f32(n)))[:, None], n + 1, axis=1).astype(f32)
pos = sc.particles.position["point"]
lo = pos.min(0); cx, cz = float(pos[:, 0].mean()), float(pos[:, 2].mean())
sc.colliders = []
sc.heightfields = [abi.heightfield(heights, (28.0, 1.5, 28.0), translation=(cx, float(lo[1]) - 2.0 * sc.cell_width + 1.5, cz), friction=0.5, below_is_solid=True)]
run("heightfield", sc)
sc = scenes.sand_column(200, 250, 200)
v, idx = scenes.box_mesh([14.0, 0.5, 14.0])
sc.colliders = []
sc.trimeshes = [abi.trimesh(v, idx, translation=(cx, float(lo[1]) - 1.5 * sc.cell_width - 0.5, cz), rotation=(0.0, 0.0, 0.0, 1.0), friction=0.5, oriented=True)]
run("trimesh box", sc)
