"""10M-particle Drucker-Prager column, 4000 substeps (1 s of collapse): throughput per 500 substeps, re-bins, reorders, C rebuilds,
the structure validator and a sanity check of the state at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
which = sys.argv[1] if len(sys.argv) > 1 else "sand"
sc = scenes.sand_column(200, 250, 200) if which == "sand" else (scenes.fluid_dam(344, 240, 244) if which == "fluid" else scenes.elastic_cube(100))
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    if os.environ.get("REORDER_HEADS"): p.reorder_policy_set(float(os.environ["REORDER_HEADS"]) * len(sc.particles), 0.0)   # excess heads per particle; 0 = never
    p.clock_enable(True, stride=8)
    for chunk in range(nchunks):
        p.clock_read(reset=True)
        p.sync(); t0 = time.perf_counter()
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 500)
        p.sync(); ms = 1e3 * (time.perf_counter() - t0) / 500
        k = p.clock_read()
        print(f"substeps {500 * chunk:4d}..{500 * chunk + 499}: {ms:.3f} ms per substep ({len(sc.particles) / ms / 1e6:.2f} G particle-substeps/s), re-bins {int(k.resorts)}, "
              f"g2p2g {k.g2p2g_ms / max(k.g2p2g_launches, 1):.3f} p2g {k.p2g_ms / max(k.p2g_launches, 1):.3f} g2p {k.g2p_ms / max(k.g2p_launches, 1):.3f} gu {k.grid_update_ms / max(k.grid_update_launches, 1):.3f} sort {k.sort_ms:.1f} reorder {k.reorder_ms:.1f} ms, heads {p.reorder_policy_stats()['last_heads']}, "
              f"blocks {p.num_particle_blocks()} / {p.num_active_blocks()}, reorders so far {p.reorder_policy_stats()['reorders']}, C rebuilt so far {p.step_stats()['c_regathers']}", flush=True)
    c = p.grid_check()
    part = p.read_particles()
    pos, v = part.position["point"], part.velocity["vector"]
    print("grid check:", c)
    print("finite:", bool(np.isfinite(pos).all() and np.isfinite(v).all()), "failed:", int(part.status["failed"].sum()), "vmax %.2f" % float(np.abs(v).max()),
          "height %.2f -> %.2f" % (float(sc.particles.position["point"][:, 1].max()), float(pos[:, 1].max())), "footprint x %.2f -> %.2f" % (float(np.ptp(sc.particles.position["point"][:, 0])), float(np.ptp(pos[:, 0]))))
