import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
for build in (lambda: scenes.sand_column(20, 12, 16, with_block=True), lambda: scenes.sand_column(12, 10, 8, with_block=True)):
    sc = build()
    p = MpmPipeline(sc.cell_width)
    p.grid_debug_capture(True)
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, 2.5e-4, 1)
    print("n", len(p), "nodes", len(p.read_grid_nodes()), "pblocks", p.num_particle_blocks(), flush=True)
    p.close()
