"""ms per substep and per-kernel spans for the non-headline BASELINE configs (neo-Hookean cube, fluid dam with the volume
recompute sweep): python tools/config_bench.py"""
import json
import sys
import time

sys.path.insert(0, ".")
from sparkl_amd import scenes                            # noqa: E402
from sparkl_amd.pipeline import MpmPipeline              # noqa: E402

for name, sc in (("neo_hookean_1M", scenes.elastic_cube(100)), ("fluid_dam_2M", scenes.fluid_dam(160, 120, 104))):
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 8)
        p.clock_enable(True, stride=2)
        p.clock_read(reset=True)
        p.sync()
        t0 = time.perf_counter()
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 64)
        p.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 64
        k = p.clock_read()
        n = len(sc.particles)
        out = {"config": name, "particles": n, "ms_per_substep": round(ms, 4), "Mparticle_substeps_per_s": round(n / ms / 1e3, 1),
               "p2g_ms": round(k.p2g_ms / max(k.p2g_launches, 1), 4), "g2p_ms": round(k.g2p_ms / max(k.g2p_launches, 1), 4),
               "g2p2g_ms": round(k.g2p2g_ms / max(k.g2p2g_launches, 1), 4), "grid_update_ms": round(k.grid_update_ms / max(k.grid_update_launches, 1), 4),
               "launches": [int(k.p2g_launches), int(k.g2p_launches), int(k.g2p2g_launches)], "resorts": int(k.resorts), "sort_ms": round(k.sort_ms, 3),
               "pblocks": p.num_particle_blocks()}
        print(json.dumps(out))
