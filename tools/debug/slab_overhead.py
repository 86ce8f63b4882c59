"""Fixed cost of the slab loop on ONE GPU: K x-slabs of a column (sparkl_hip_group_substeps, device-copy transport of the loop that
uses RCCL between ranks) against ONE pipeline holding the same column.  python tools/debug/slab_overhead.py [K=2] [particles per slab=10e6] [substeps=100]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from sparkl_amd.slab import INT32_MAX, INT32_MIN
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = float(sys.argv[2]) if len(sys.argv) > 2 else 10e6
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 100
f = (per / 10e6) ** (1 / 3)
nx, ny, nz = max(8, int(round(200 * f / 8)) * 8), int(round(250 * f)), int(round(200 * f))
stream = torch.cuda.Stream()
def timed(fn, n):
    fn(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
# one pipeline, the whole column
sc = scenes.sand_column(K * nx, ny, nz)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream) as p:
    p.load_scene(sc)
    single = timed(lambda n: (p.substeps(sc.params, sc.gravity, sc.fixed_dt, n), p.sync()), nsub)
    nb = p.num_particle_blocks()
print(f"one pipeline, {len(sc.particles)} particles ({K * nx}x{ny}x{nz}), {nb} particle blocks: {single:.4f} ms per substep", flush=True)
params, gravity, dt, h, bits = sc.params, sc.gravity, sc.fixed_dt, sc.cell_width, sc.domain_log2_bytes
del sc
pipes = []
for r in range(K):
    s = scenes.sand_column(nx, ny, nz, x_slab=(r, K))
    blocks = nx // 8
    p = MpmPipeline(h, device_id=0, domain_log2_bytes=s.domain_log2_bytes, stream=stream.cuda_stream)
    p.set_slab(0, INT32_MIN if r == 0 else 1 + r * blocks, INT32_MAX if r == K - 1 else 1 + (r + 1) * blocks, id_offset=r * len(s.particles))
    p.set_models(s.models); p.set_colliders(s.colliders); p.upload_particles(s.particles)
    pipes.append(p)
    del s
pipes[0].clock_enable(True, 4)
slabs = timed(lambda n: MpmPipeline.group_substeps(pipes, params, gravity, dt, n), nsub)
c = pipes[0].clock_read(True)
print(f"{K} slabs of {nx}x{ny}x{nz} through sparkl_hip_group_substeps: {slabs:.4f} ms per substep = {slabs / single:.3f} x the single pipeline; "
      f"overhead {1e3 * (slabs - single) / K:.1f} us per slab-substep; member 0: halo spans {c.halo_ms / max(c.sampled_substeps, 1):.4f} ms per substep, "
      f"g2p2g {c.g2p2g_ms / max(c.g2p2g_launches, 1):.4f}, grid {c.grid_update_ms / max(c.grid_update_launches, 1):.4f}, re-bins {[q.group_rebins() for q in pipes]}")
for p in pipes: p.close()
