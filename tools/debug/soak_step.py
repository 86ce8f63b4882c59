"""MpmPipeline::step (dt = 1/60) on the 10M column for one second of collapse: ms per frame, substeps, look-ahead statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(200, 250, 200)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    tsim = 0.0
    for block in range(6):
        nsub = 0
        p.sync(); t0 = time.perf_counter()
        for _ in range(10):
            t, tim = p.step(sc.params, sc.gravity, timings=True)
            tsim += t; nsub += tim.num_substeps
        p.sync(); ms = 1e3 * (time.perf_counter() - t0) / 10
        print(f"frames {10 * block:2d}..{10 * block + 9}: {ms:.2f} ms per frame, {nsub / 10:.1f} substeps per frame ({ms * 10 / nsub:.3f} ms per substep), t = {tsim:.3f} s, stats {p.step_stats()}", flush=True)
    c = p.grid_check()
    v = p.read_velocities()["vector"]
    print("grid check ok:", all(c[k] == 0 for k in list(c)[:5]), "finite:", bool(np.isfinite(v).all()), "failed:", int(p.read_statuses()["failed"].sum()))
