import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
sc = scenes.sand_column(16, 10, 12, with_block=True)
dt = 2.5e-4
res = {}
for name, f64 in (("f64", True), ("f32", False)):
    o = Oracle(sc.cell_width, f64=f64, domain_log2_bytes=sc.domain_log2_bytes, threads=8); o.load_scene(sc)
    for _ in range(10): o.substep_fixed(sc.params, sc.gravity, dt)
    res[name] = o.get_particles(); o.close()
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc); p.substeps(sc.params, sc.gravity, dt, 10); res["hip"] = p.read_particles()
ref = res["f64"]
vmax = np.abs(ref.velocity["vector"]).max()
for name in ("hip", "f32"):
    d = np.linalg.norm(res[name].velocity["vector"].astype(np.float64) - ref.velocity["vector"].astype(np.float64), axis=1) / vmax
    idx = np.argsort(-d)[:8]
    y = ref.position["point"][:, 1]
    print(name, "max", d.max(), "worst y:", np.round(y[idx], 3), "err", np.round(d[idx] * 1e4, 2), "model", ref.status["model_index"][idx])
    ys = np.unique(np.round(y, 1))
    print("   mean err by height band:", {float(b): float(np.round(d[np.abs(y - b) < 0.051].mean() * 1e4, 3)) for b in ys[:6]}, "... top", {float(b): float(np.round(d[np.abs(y - b) < 0.051].mean() * 1e4, 3)) for b in ys[-3:]})
print("floor top y:", 4 * sc.cell_width - 0.5 * sc.cell_width, "h", sc.cell_width)
