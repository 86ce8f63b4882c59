"""More long runs: adaptive frames on the 20M fluid dam (volume recompute + estimator every substep); adaptive frames on two slabs
(sparkl_hip_group_step); the 10M column on the sand3 heightfield ground."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sparkl_amd import scenes, abi
from sparkl_amd.pipeline import MpmPipeline
from sparkl_amd.particles import ParticleComponents
from sparkl_amd.slab import block_cuts_balanced
f32 = np.float32
which = sys.argv[1]
if which == "fluid_step":
    sc = scenes.fluid_dam(344, 240, 244)
    sc.params.dt = 2.0e-3
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        for block in range(4):
            nsub = 0; p.sync(); t0 = time.perf_counter()
            for _ in range(10):
                _, tim = p.step(sc.params, sc.gravity, timings=True); nsub += tim.num_substeps
            p.sync(); ms = 1e3 * (time.perf_counter() - t0) / 10
            print(f"fluid frames {10 * block}..{10 * block + 9}: {ms:.2f} ms per frame, {nsub / 10:.1f} substeps per frame, {ms * 10 / nsub:.3f} ms per substep", flush=True)
        c = p.grid_check(); v = p.read_velocities()["vector"]
        print("validator ok:", all(c[k] == 0 for k in list(c)[:5]), "finite:", bool(np.isfinite(v).all()), "failed:", int(p.read_statuses()["failed"].sum()))
elif which == "group_step":
    sc = scenes.sand_column(120, 100, 60)
    h = sc.cell_width
    cell = np.round(sc.particles.position["point"][:, 0] / f32(h)).astype(np.int64) - 1
    cuts = block_cuts_balanced(cell >> 2, 2)
    gids = [np.nonzero(((cell >> 2) >= cuts[r]) & ((cell >> 2) < cuts[r + 1]))[0] for r in range(2)]
    stream = torch.cuda.Stream(); pipes = []
    for r in range(2):
        comps = ParticleComponents(0)
        for k in ParticleComponents.__slots__: setattr(comps, k, np.ascontiguousarray(getattr(sc.particles, k)[gids[r]]))
        p = MpmPipeline(h, device_id=0, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
        p.set_slab(0, cuts[r], cuts[r + 1], id_offset=int(sum(len(g) for g in gids[:r])))
        p.set_models(sc.models); p.set_colliders(sc.colliders); p.upload_particles(comps); pipes.append(p)
    tsim = 0.0
    for block in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); nsub = 0
        for _ in range(10):
            t, n = MpmPipeline.group_step(pipes, sc.params, sc.gravity); tsim += t; nsub += n
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 10
        print(f"two slabs, frames {10 * block}..{10 * block + 9}: {ms:.2f} ms per frame, {nsub / 10:.1f} substeps, t = {tsim:.3f} s, per slab {[len(p) for p in pipes]}, re-bins {[p.group_rebins() for p in pipes]}", flush=True)
    ids = np.concatenate([p.read_sorted_particle_ids() for p in pipes])
    print("ids complete:", bool(np.array_equal(np.sort(ids), np.arange(len(sc.particles)))), "validators:", [sum(list(p.grid_check().values())[1:5]) for p in pipes])
    for p in pipes: p.close()
