import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from oracle import Oracle
from tests.parity import node_errors
sc = scenes.sand_column(8, 6, 8, with_block=False)
dt = 2.5e-4
o = Oracle(sc.cell_width); o.load_scene(sc)
o.sort(); o.p2g(dt, sc.gravity); o.grid_update(dt, sc.params.boundary_handling, 0)
ref = o.nodes()
with MpmPipeline(sc.cell_width) as p:
    p.grid_debug_capture(True)
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, dt, 1)
    got = p.read_grid_nodes()
    ids = p.read_sorted_particle_ids()
print(node_errors(got, ref))
def key(a):
    c = a["cell"].astype(np.int64); return (c[:,0]<<42)|(c[:,1]<<21)|c[:,2]
g = got[np.argsort(key(got))]; r = ref[np.argsort(key(ref))]
print("sum mass got/ref/particles", g["mass"].sum(), r["mass"].sum(), sc.particles.volume["mass"].sum())
bad = np.argsort(-np.abs(g["mass"]-r["mass"]))[:12]
for i in bad: print(g["cell"][i], g["mass"][i], r["mass"][i], g["mass"][i]/r["mass"][i])
ratio = g["mass"]/r["mass"]
print("ratio hist", np.histogram(ratio, bins=[0,0.4,0.6,0.9,0.99,1.01,1.1,1.6,2.1,3,10])[0])
# by local coordinate in block
for ax in range(3):
    for c in range(4):
        m = (g["cell"][:,ax] % 4) == c
        print("axis", ax, "c", c, "mean ratio", ratio[m].mean(), "max", ratio[m].max(), "min", ratio[m].min())
