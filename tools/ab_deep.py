"""A/B of library variants in three phases of the 10M collapse on one box: python tools/ab_deep.py name=lib.so ...  (empty lib = working tree; name=lib.so,ENV=VAL sets environment variables for that variant).
Each variant runs the column once: 64 timed substeps from substep 100, 640 and 2000 (kernel spans from the library clock)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
sc = scenes.sand_column(200, 250, 200)
out = []
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    done = 0
    for mark in (100, 640, 2000):
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, mark - done); done = mark
        p.clock_enable(True, 2); p.clock_read(True); p.sync()
        t0 = time.perf_counter(); p.substeps(sc.params, sc.gravity, sc.fixed_dt, 128); p.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 128; done += 128
        c = p.clock_read(True); p.clock_enable(False)
        out.append("%%d: %%.4f (g2p2g %%.4f grid %%.4f)" %% (mark, ms, c.g2p2g_ms / max(c.g2p2g_launches, 1), c.grid_update_ms / max(c.grid_update_launches, 1)))
print(" | ".join(out))
''' % root
for rep in range(2):
    for spec in sys.argv[1:]:
        name, rest = spec.split("=", 1)
        lib, *kvs = rest.split(",")
        env = dict(os.environ)
        if lib: env["SPARKL_HIP_LIB"] = os.path.join(root, lib)
        for kv in kvs:
            k, v = kv.split("=")
            env[k] = v
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"{name:10s}", (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1], flush=True)
