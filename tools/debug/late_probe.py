"""ms per substep of the 10M column over substeps 2000..2400 (deep in the flowing phase) for a library variant (SPARKL_HIP_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(200, 250, 200)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    if os.environ.get("REORDER_HEADS"): p.reorder_policy_set(float(os.environ["REORDER_HEADS"]) * len(sc.particles), 0.0)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 2000)
    p.sync(); t0 = time.perf_counter()
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 400)
    p.sync(); ms = 1e3 * (time.perf_counter() - t0) / 400
    print(f"{os.environ.get('SPARKL_HIP_LIB', 'default')} heads {os.environ.get('REORDER_HEADS', 'default')}: {ms:.3f} ms per substep, reorders {p.reorder_policy_stats()['reorders']}")
