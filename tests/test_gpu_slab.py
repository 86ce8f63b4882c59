"""Two slabs on ONE GPU (two processes, host-staged gloo exchange): the real HIP halo pack / merge / migration kernels
driven by the same SlabStepper the multi-GPU bench uses with RCCL.  Result must match the un-decomposed pipeline."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sparkl_amd import scenes  # noqa: E402
from sparkl_amd.particles import ParticleComponents  # noqa: E402
from sparkl_amd.slab import HipSlabBackend, SlabStepper, block_cuts_balanced  # noqa: E402

pytestmark = pytest.mark.gpu


def _scene(kind):
    if kind == "spin":
        sc = scenes.elastic_cube(20)
        sc.particles.velocity["vector"] *= 25.0
        return sc, 2.0e-4
    sc = scenes.sand_column(24, 12, 16, with_block=True)
    return sc, 2.5e-4


def _worker(rank, world, port, kind, nsteps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparkl_amd.pipeline import MpmPipeline
    sc, dt = _scene(kind)
    h = sc.cell_width
    cell = np.round(sc.particles.position["point"][:, 0] / np.float32(h)).astype(np.int64) - 1
    cuts = block_cuts_balanced(cell >> 2, world)
    mine = ((cell >> 2) >= cuts[rank]) & ((cell >> 2) < cuts[rank + 1])
    # contiguous id ranges per rank: remap ids so that rank r uploads ids [offset, offset + n)
    order = np.argsort(~mine, kind="stable") if rank == 0 else np.argsort(mine, kind="stable")
    comps = ParticleComponents(0)
    for k in ParticleComponents.__slots__:
        setattr(comps, k, np.ascontiguousarray(getattr(sc.particles, k)[mine]))
    gids = np.nonzero(mine)[0]
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        pipe = MpmPipeline(h, device_id=0, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
        pipe.set_slab(0, cuts[rank], cuts[rank + 1], id_offset=0)
        pipe.set_models(sc.models)
        pipe.set_colliders(sc.colliders)
        pipe.upload_particles(comps)
        st = SlabStepper(HipSlabBackend(pipe, torch.device("cuda", 0)), rank, world, chunk=8)
        dts = []
        if nsteps > 0:
            st.substeps(sc.params, sc.gravity, dt, nsteps)
        else:   # adaptive: -nsteps frames of params.dt through SlabStepper.step
            for _ in range(-nsteps):
                dts += st.step(sc.params, sc.gravity)[1]
        got = pipe.read_particles()
        local_ids = pipe.read_sorted_particle_ids()
    # local ids index this rank's upload order only for particles that never migrated; carry global ids through `user`:
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pos=got.position["point"], vel=got.velocity["vector"], F=got.volume["deformation_gradient"],
             mass=got.volume["mass"], rebins=st.num_rebins, n0=len(gids), n1=len(got), dts=np.array(dts))
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _match(pos_a, pos_b):
    """order-free matching of two particle sets through a lexicographic sort of (rounded) positions"""
    ka = np.lexsort(np.round(pos_a.astype(np.float64) * 1e4).T)
    kb = np.lexsort(np.round(pos_b.astype(np.float64) * 1e4).T)
    return ka, kb


@pytest.mark.parametrize("kind,nsteps", [("spin", 40), ("sand", 30)])
def test_two_slabs_one_gpu_match_single_pipeline(tmp_path, kind, nsteps):
    from sparkl_amd.pipeline import MpmPipeline
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, nsteps, str(tmp_path)), nprocs=world, join=True)
    sc, dt = _scene(kind)
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, nsteps)
        ref = p.read_particles()
    parts = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    pos = np.concatenate([q["pos"] for q in parts])
    vel = np.concatenate([q["vel"] for q in parts])
    F = np.concatenate([q["F"] for q in parts])
    assert len(pos) == len(sc.particles), "particles lost or duplicated across slabs"
    ka, kb = _match(pos, ref.position["point"])
    h = sc.cell_width
    vmax = np.abs(ref.velocity["vector"]).max()
    ex = np.abs(pos[ka] - ref.position["point"][kb]).max() / h
    ev = np.abs(vel[ka] - ref.velocity["vector"][kb]).max() / vmax
    eF = np.abs(F[ka] - ref.volume["deformation_gradient"][kb]).max()
    print(kind, "slab vs single:", ex, ev, eF, [int(q["rebins"]) for q in parts], [(int(q["n0"]), int(q["n1"])) for q in parts])
    assert ex < 2e-5 and ev < 2e-4 and eF < 2e-5
    if kind == "spin":
        assert any(int(q["n0"]) != int(q["n1"]) for q in parts), "the spinning cube must migrate particles across the cut"


def test_two_slabs_adaptive_step_matches_single_pipeline(tmp_path):
    """`SlabStepper.step` (dt bound = MIN over the ranks) against `sparkl_hip_pipeline_step` on the un-decomposed set."""
    from sparkl_amd.pipeline import MpmPipeline
    world, frames = 2, 2
    mp.spawn(_worker, args=(world, _free_port(), "sand", -frames, str(tmp_path)), nprocs=world, join=True)
    sc, _ = _scene("sand")
    nsub = 0
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        for _ in range(frames):
            _, tim = p.step(sc.params, sc.gravity, timings=True)
            nsub += tim.num_substeps
        ref = p.read_particles()
    parts = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    for q in parts:
        assert len(q["dts"]) == nsub and abs(float(q["dts"].sum()) - frames * sc.params.dt) < 1e-6
    assert np.array_equal(parts[0]["dts"], parts[1]["dts"])
    pos = np.concatenate([q["pos"] for q in parts])
    vel = np.concatenate([q["vel"] for q in parts])
    assert len(pos) == len(sc.particles)
    ka, kb = _match(pos, ref.position["point"])
    ex = np.abs(pos[ka] - ref.position["point"][kb]).max() / sc.cell_width
    ev = np.abs(vel[ka] - ref.velocity["vector"][kb]).max() / np.abs(ref.velocity["vector"]).max()
    print("adaptive slab vs single:", ex, ev, nsub)
    assert ex < 2e-5 and ev < 2e-4
