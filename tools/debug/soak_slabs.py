"""cube_through_sand (configs[4]'s scene, 1.7M particles) on FOUR slabs that share one GPU and one stream, stepped by
sparkl_hip_group_substeps (device-copy transport of the loop that uses RCCL between ranks): 1000 substeps, the kinematic cube ploughs
through three cuts.  Throughput, collective re-bins, migration, re-balancing, particle conservation, per-slab structure check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
from sparkl_amd.particles import ParticleComponents
from sparkl_amd.slab import block_cuts_balanced
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = scenes.cube_through_sand(240, 60, 120)
if len(sys.argv) > 2:      # drift: the whole bed moves along x, so particles migrate across every cut
    sc.particles.velocity["vector"][:, 0] += np.float32(sys.argv[2])
h = sc.cell_width
cell = np.round(sc.particles.position["point"][:, 0] / np.float32(h)).astype(np.int64) - 1
cuts = block_cuts_balanced(cell >> 2, world)
gids = [np.nonzero(((cell >> 2) >= cuts[r]) & ((cell >> 2) < cuts[r + 1]))[0] for r in range(world)]
stream = torch.cuda.Stream()
pipes = []
for r in range(world):
    comps = ParticleComponents(0)
    for k in ParticleComponents.__slots__:
        setattr(comps, k, np.ascontiguousarray(getattr(sc.particles, k)[gids[r]]))
    p = MpmPipeline(h, device_id=0, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
    p.set_slab(0, cuts[r], cuts[r + 1], id_offset=int(sum(len(g) for g in gids[:r])))
    p.set_models(sc.models); p.set_colliders(sc.colliders); p.upload_particles(comps)
    pipes.append(p)
n0 = [len(g) for g in gids]
print(len(sc.particles), "particles on", world, "slabs:", n0, "cuts", [int(c) for c in cuts])
for block in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    MpmPipeline.group_substeps(pipes, sc.params, sc.gravity, sc.fixed_dt, 200)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 200
    n = [len(p) for p in pipes]
    chk = [p.grid_check() for p in pipes]
    bad = [sum(c[k] for k in ("blocks_not_in_hash", "particles_outside_their_block", "missing_neighbour_blocks", "coverage_errors")) for c in chk]
    print(f"substeps {200 * block:4d}..{200 * block + 199}: {ms:.3f} ms per substep ({len(sc.particles) / ms / 1e6:.2f} G particle-substeps/s), per slab {n} (sum {sum(n)}), "
          f"collective re-bins {[p.group_rebins() for p in pipes]}, cuts {[p.slab_get()[1:3] if hasattr(p, 'slab_get') else None for p in pipes]}, validator violations {bad}", flush=True)
ids = np.concatenate([p.read_sorted_particle_ids() for p in pipes])
print("ids unique and complete:", bool(np.array_equal(np.sort(ids), np.arange(len(sc.particles)))))
for p in pipes: p.close()
