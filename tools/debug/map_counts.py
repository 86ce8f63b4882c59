"""Outcomes of the Drucker-Prager return map per wave along the 10M collapse (analysis build, tools/debug/ablate_build.sh):
python tools/debug/map_counts.py [marks ...]   -> per window of substeps: share of waves whose lanes are ALL tensile / hold a tensile lane /
hold no lane that moves, and the share of lanes that move (tensile or onto the cone)."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPARKL_HIP_LIB"] = os.path.join(root, "sparkl_amd/lib/ab_ablate.so"); os.environ["SPARKL_HIP_LIB_AB"] = "1"
sys.path.insert(0, root)
from sparkl_amd import scenes, _lib
from sparkl_amd.pipeline import MpmPipeline
marks = [int(a) for a in sys.argv[1:]] or [5, 25, 45, 120, 640, 768, 2000, 2128]
sc = scenes.sand_column(200, 250, 200)
L = _lib.lib()
out = (C.c_ulonglong * 8)()
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    done = 0
    for m in marks:
        L.sparkl_hip_debug_map_counts(out, 1)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, m - done); p.sync()
        L.sparkl_hip_debug_map_counts(out, 0)
        w, allt, anyt, nomove, moved, tens, lanes = [float(out[i]) for i in range(7)]
        print(f"substeps {done:5d}..{m:5d}: waves all tensile {allt / max(w, 1):.3f}, any tensile {anyt / max(w, 1):.3f}, none moving {nomove / max(w, 1):.3f}; lanes moving {moved / max(lanes, 1):.3f}, tensile {tens / max(lanes, 1):.3f}")
        done = m
