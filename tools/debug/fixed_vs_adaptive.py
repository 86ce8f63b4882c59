"""k_g2p2g per launch at the adaptive step length (1/360 s) but through the fixed-dt batch, next to MpmPipeline::step:
separates the cost of the regime (larger displacements per substep) from the cost of the adaptive machinery."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(200, 250, 200)
dt = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0 / 360.0
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, dt, 6)
    p.clock_enable(1)
    p.sync(); t0 = time.perf_counter()
    p.substeps(sc.params, sc.gravity, dt, 96)
    p.sync(); ms = 1e3 * (time.perf_counter() - t0)
    c = p.clock_read()
    print(f"fixed dt={dt:.3e}: {ms / 96:.3f} ms per substep; clock: " + " ".join(f"{k}={getattr(c, k):.3f}" for k in ("p2g_ms", "grid_update_ms", "g2p_ms", "g2p2g_ms", "sort_ms", "reorder_ms") if hasattr(c, k)) + f" launches g2p2g={c.g2p2g_launches} g2p={c.g2p_launches} resorts={c.resorts}")
