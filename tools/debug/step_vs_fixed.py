"""MpmPipeline::step (adaptive, dt = 1/60 frames, 6 substeps of 1/360 each on the E = 1e7 column) against the fixed-dt batch at the SAME
step length and the SAME phase of the collapse: frames 0..9, 30..39 and 50..59 of one run each.  python tools/debug/step_vs_fixed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(200, 250, 200)
sc.params.dt = 1.0 / 60.0
res = {}
for mode in ("step", "fixed"):
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        out, nsub_total = [], 0
        for frame in range(60):
            p.sync(); t0 = time.perf_counter()
            if mode == "step":
                p.step(sc.params, sc.gravity)
                n = 6
            else:
                p.substeps(sc.params, sc.gravity, sc.params.dt / 6.0, 6)
                n = 6
            p.sync(); out.append(1e3 * (time.perf_counter() - t0) / n)
        res[mode] = out
        st = p.step_stats() if mode == "step" else None
    print(mode, "first 12 frames, ms per frame:", " ".join(f"{6 * x:.2f}" for x in out[:12]))
    print(mode, "ms per substep, frames 0-9 / 30-39 / 50-59:", " / ".join(f"{sum(out[a:a + 10]) / 10:.3f}" for a in (0, 30, 50)), st or "", flush=True)
for a in (0, 30, 50):
    s, f = sum(res["step"][a:a + 10]) / 10, sum(res["fixed"][a:a + 10]) / 10
    print(f"frames {a}-{a + 9}: step / fixed = {s / f:.3f}")
