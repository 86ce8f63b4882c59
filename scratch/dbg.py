import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(20, 12, 16, with_block=True)
p = MpmPipeline(sc.cell_width)
print("created", flush=True)
p.grid_debug_capture(True); print("capture", flush=True)
p.load_scene(sc); print("loaded", flush=True)
p.substeps(sc.params, sc.gravity, 2.5e-4, 1); print("stepped", flush=True)
p.sync(); print("synced", flush=True)
n = p.read_grid_nodes(); print("nodes", len(n), flush=True)
c = p.read_particles(); print("read", flush=True)
print("len", len(p), "pblocks", p.num_particle_blocks(), "blocks", p.num_active_blocks(), flush=True)
ids = p.read_sorted_particle_ids(); print("ids", ids[:8], ids.max(), flush=True)
print("pos", c.position["point"][:3], sc.particles.position["point"][:3])
st = p.read_statuses(); print("failed", st["failed"].sum())
