import os, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from sparkl_amd._lib import check
from oracle import Oracle
from tests.parity import node_errors
sc = scenes.sand_column(20, 12, 16, with_block=True)
dt = 2.5e-4
o = Oracle(sc.cell_width); o.load_scene(sc); o.sort(); o.p2g(dt, sc.gravity); o.grid_update(dt, sc.params.boundary_handling, 0)
ref = o.nodes()
p = MpmPipeline(sc.cell_width)
p.grid_debug_capture(True)
p.load_scene(sc)
check(p.L.sparkl_hip_rebin(p.h))
g = (C.c_float * 3)(*sc.gravity)
check(p.L.sparkl_hip_substep_begin(p.h, C.byref(sc.params), g, C.c_float(dt)))
check(p.L.sparkl_hip_substep_finish(p.h))
p.sync()
got = p.read_grid_nodes()
print("nodes got/ref", len(got), len(ref))
print(node_errors(got, ref))
print("max |v| got", np.abs(got["velocity"]).max(), "ref", np.abs(ref["velocity"]).max(), "mass sum", got["mass"].sum(), ref["mass"].sum())
bad = np.argsort(-np.abs(got["velocity"]).max(1))[:5]
print(got[bad])
