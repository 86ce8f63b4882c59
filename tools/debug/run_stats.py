"""Run structure the fused scatter sees along the 10M collapse (sparkl_hip_debug_run_stats, probes.h) with the kernel times beside it:
python tools/debug/run_stats.py [substep marks ...]   (default 0 640 2000)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparkl_amd import scenes, _lib
from sparkl_amd.pipeline import MpmPipeline
marks = [int(a) for a in sys.argv[1:]] or [0, 640, 2000]
NAMES = ["nruns", "maxlen", "distinct", "chunks4", "rounds", "iters", "pcount/64", "live", "rows", "layers", "|S|<1e-3,3e-3,1e-2,3e-2,.1,.3,inf", "maxcell", "iters_multi"]
def stats(p):
    out = (C.c_ulonglong * (13 * 65))()
    _lib.check(_lib.lib().sparkl_hip_debug_run_stats(p.h, out, C.c_uint64(len(out))))
    a = np.array(out, dtype=np.float64).reshape(13, 65)
    for k, name in enumerate(NAMES):
        h = a[k]; tot = h.sum()
        if tot == 0: continue
        mean = (h * np.arange(65)).sum() / tot
        cum = np.cumsum(h) / tot
        q = [int(np.searchsorted(cum, x)) for x in (0.1, 0.5, 0.9, 0.99)]
        top = " ".join(f"{i}:{h[i] / tot:.3f}" for i in np.argsort(-h)[:8] if h[i] > 0)
        print(f"  {name:10s} n {int(tot):8d} mean {mean:6.2f} p10/50/90/99 {q}  top {top}")
sc = scenes.sand_column(200, 250, 200)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    done = 0
    for m in marks:
        if m > done: p.substeps(sc.params, sc.gravity, sc.fixed_dt, m - done); done = m
        p.sync()
        print(f"== after {done} substeps: pblocks {p.num_particle_blocks()} blocks {p.num_active_blocks()}")
        stats(p)
        p.clock_enable(True, 1); p.clock_read(True)
        t0 = time.perf_counter()
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 64); p.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 64
        done += 64
        c = p.clock_read(True); p.clock_enable(False)
        print(f"  64 substeps: {ms:.3f} ms/substep; g2p2g {c.g2p2g_ms / max(c.g2p2g_launches, 1):.4f} ms x {c.g2p2g_launches}, grid {c.grid_update_ms / max(c.grid_update_launches, 1):.4f}, reorder {c.reorder_ms:.2f} ms total, resorts {c.resorts}")
        print("  after them:"); stats(p)
