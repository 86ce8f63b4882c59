"""Where k_g2p2g's time goes in a given phase of the 10M collapse: an ANALYSIS build of the library (tools/ablate_build.sh:
a copy of csrc/ with runtime switches patched in, never the product) runs the column to a mark with everything on, then flips one
switch and times the next few launches.  The switched-off part leaves wrong physics behind, so every (mark, mode) is its own process.
    python tools/ablate.py [mark ...]       (needs sparkl_amd/lib/ab_ablate.so)
modes (bit mask): 1 no deposit atomics | 2 deposit to lane-private slots (no address or bank conflicts) | 4 one visit per task |
                  8 gather without its LDS reads | 16 no scatter at all | 64 return map always on its series path |
                  128 no return map and no stress | 256 no positive-energy bookkeeping | 512 no gather (v = 0, C = 0)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
from sparkl_amd import scenes, _lib
from sparkl_amd.pipeline import MpmPipeline
mark, mode, nt = int(sys.argv[1]), int(sys.argv[2]), 3
sc = scenes.sand_column(200, 250, 200)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, mark); p.sync()
    assert _lib.lib().sparkl_hip_debug_set_ablate(mode) == 0
    p.clock_enable(True, 2); p.clock_read(True); p.sync()
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, nt); p.sync()
    c = p.clock_read(True)
    print("%%.4f %%d %%d" %% (c.g2p2g_ms / max(c.g2p2g_launches, 1), c.g2p2g_launches, c.noop_launches))
''' % root
env = dict(os.environ, SPARKL_HIP_LIB=os.path.join(root, "sparkl_amd/lib/ab_ablate.so"), SPARKL_HIP_LIB_AB="1")
marks = [int(a) for a in sys.argv[1:]] or [100, 704, 2064]
MODES = [(0, "all on"), (1, "no deposit"), (2, "private deposit"), (4, "one visit"), (5, "one visit, no deposit"), (16, "no scatter"), (8, "no gather reads"), (24, "no gather reads, no scatter"), (64, "series path forced"), (128, "no return map, no stress"), (256, "no psi_pos"), (512, "no gather"), (912, "no gather / return map / psi / scatter")]
if os.environ.get("ABLATE_MODES"):   # e.g. ABLATE_MODES=0,64,80
    MODES = [(int(m), dict(MODES).get(int(m), "mode " + m)) for m in os.environ["ABLATE_MODES"].split(",")]
for mark in marks:
    for mode, name in MODES:
        r = subprocess.run([sys.executable, "-c", CODE, str(mark), str(mode)], env=env, capture_output=True, text=True)
        print(f"mark {mark:5d}  {name:28s}", (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1], flush=True)
