"""Per-frame cost over a long collapse: ms per substep, re-bins and their cost (10M-particle column by default)."""
import json
import sys
import time

sys.path.insert(0, ".")
from sparkl_amd import scenes                            # noqa: E402
from sparkl_amd.pipeline import MpmPipeline              # noqa: E402

n_target = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0e6
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 24
side = int(round((n_target / 1.25) ** (1.0 / 3.0) / 8.0)) * 8
sc = scenes.sand_column(side, int(side * 1.25), side)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    if len(sys.argv) > 3:
        p.reorder_policy_set(float(sys.argv[3]), float(sys.argv[4]) if len(sys.argv) > 4 else 0.0)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 8)
    p.clock_enable(True, stride=8)
    rows = []
    for f in range(frames):
        p.clock_read(reset=True)
        p.sync()
        t0 = time.perf_counter()
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 64)
        p.sync()
        ms = 1e3 * (time.perf_counter() - t0)
        k = p.clock_read(reset=False)
        rows.append({"frame": f, "ms_per_substep": ms / 64, "resorts": int(k.resorts), "sort_ms": round(k.sort_ms, 3),
                     "g2p2g_ms": round(k.g2p2g_ms / max(k.g2p2g_launches, 1), 4), "p2g_launches": int(k.p2g_launches), "g2p_launches": int(k.g2p_launches),
                     "pblocks": p.num_particle_blocks(), "blocks": p.num_active_blocks(), "reorder_ms": round(k.reorder_ms, 3), **p.reorder_policy_stats()})
        print(json.dumps(rows[-1]))
