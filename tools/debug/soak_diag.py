import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(200, 250, 200)
start = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, start)
    p.clock_enable(True, stride=64)
    for chunk in range(12):
        p.clock_read(reset=True)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 40)
        k = p.clock_read()
        st = p.read_statuses()
        pos = p.read_positions()["point"]; v = p.read_velocities()["vector"]
        F = p.read_volumes()["deformation_gradient"]
        failed = st["failed"] == 1
        sp = np.linalg.norm(v, axis=1)
        i = int(np.argmax(sp))
        print(f"substep {start + 40 * (chunk + 1)}: re-bins {int(k.resorts)} failed {int(failed.sum())} vmax {sp.max():.2f} at {pos[i]} (failed {bool(failed[i])}) |F-I|max over unfailed {np.abs(F[~failed] - np.eye(3, dtype=np.float32).reshape(9)).max():.3f} ymin {pos[:,1].min():.2f} xmin {pos[:,0].min():.2f} xmax {pos[:,0].max():.2f}", flush=True)
