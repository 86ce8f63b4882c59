"""ms per adaptive frame (MpmPipeline::step, dt = 1/60) on the 10M sand column and the 20M fluid dam."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
which = sys.argv[1] if len(sys.argv) > 1 else "sand"
sc = scenes.sand_column(200, 250, 200) if which == "sand" else scenes.fluid_dam(344, 240, 244)
if which != "sand":
    sc.params.dt = 2.0e-3
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    p.step(sc.params, sc.gravity)
    n, nsub = 8, 0
    p.sync(); t0 = time.perf_counter()
    for _ in range(n):
        p.step(sc.params, sc.gravity)
    p.sync(); ms_plain = 1e3 * (time.perf_counter() - t0) / n
    p.sync(); t0 = time.perf_counter()
    for _ in range(n):
        _, tim = p.step(sc.params, sc.gravity, timings=True)
        nsub += tim.num_substeps
    p.sync(); ms = 1e3 * (time.perf_counter() - t0)
    print(f"{which}: {ms_plain:.3f} ms per frame without timings")
    print(f"{which}: {ms / n:.3f} ms per frame, {nsub / n:.1f} substeps per frame, {ms / nsub:.3f} ms per substep; last frame: p2g {tim.p2g_ms:.3f} grid {tim.grid_update_ms:.3f} g2p {tim.g2p_ms:.3f} dt {tim.estimate_timestep_ms:.3f} total {tim.total_ms:.3f}")
