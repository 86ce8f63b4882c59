"""world_size-2 gloo test of the slab orchestration (sparkl_amd/slab.py) on CPU.

The HIP backend cannot run here, so the *same* SlabStepper code drives a checker backend built on the oracle: every rank
owns the particles of one slab, packs the (mass, momentum) of the node blocks in the two planes at the cut, swaps them
with its neighbour through torch.distributed, and migrates particles at re-bins.  The merged result must match the
single-domain oracle.  This pins the protocol the GPU path uses (flag all-reduce, halo swap, emigrants/immigrants).
"""
import ctypes
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sparkl_amd import abi, scenes  # noqa: E402
from sparkl_amd.particles import ParticleComponents  # noqa: E402
from sparkl_amd.slab import INT32_MAX, INT32_MIN, SlabStepper, block_cuts_balanced, block_cuts_even  # noqa: E402

REC = np.dtype([("status", abi.STATUS_DTYPE), ("position", abi.POSITION_DTYPE), ("velocity", abi.VELOCITY_DTYPE), ("volume", abi.VOLUME_DTYPE),
                ("phase", abi.PHASE_DTYPE), ("data", abi.DATA_DTYPE), ("velocity_gradient", abi.VELGRAD_DTYPE), ("gid", "<u8")])
HALO = np.dtype([("key", "<u8"), ("pad", "<u8"), ("nodes", "<f4", (64, 4))])


class OracleSlabBackend:
    """Checker backend: the oracle computes, this class only slices by slab (exact binning, no tile margin)."""
    halo_record_bytes = HALO.itemsize
    migration_record_bytes = REC.itemsize

    def __init__(self, scene, comps, gids, axis, lo, hi):
        from oracle import Oracle
        self.o = Oracle(scene.cell_width, domain_log2_bytes=scene.domain_log2_bytes)
        self.o.set_models(scene.models)
        self.o.set_cuboids(scene.colliders)
        self.h, self.axis, self.lo, self.hi = scene.cell_width, axis, lo, hi
        self.comps, self.gids = comps, gids.astype(np.uint64)
        self.emig = [np.zeros(0, np.int64), np.zeros(0, np.int64)]
        self._drop = []
        self.next_flag = self.cur_flag = 0
        self.steps_done = 0
        self.halo_ids = [np.zeros(0, np.uint64), np.zeros(0, np.uint64)]
        self.params = self.g = self.dt = None

    def new_buffer(self, nbytes):
        return torch.zeros(max(int(nbytes), 16), dtype=torch.uint8)

    def _block_index(self):
        x = self.comps.position["point"][:, self.axis]
        cell = np.round(x / np.float32(self.h)).astype(np.int64) - 1  # associated cell (sp_grid.rs:216-226)
        return cell >> 2

    def rebin(self):
        from oracle import lib
        b = self._block_index()
        self.emig = [np.nonzero(b < self.lo)[0], np.nonzero(b >= self.hi)[0]]
        keep = np.ones(len(b), bool)
        keep[self.emig[0]] = False
        keep[self.emig[1]] = False
        local = ParticleComponents(0)
        for k in ParticleComponents.__slots__:
            setattr(local, k, np.ascontiguousarray(getattr(self.comps, k)[keep]))
        self.local_idx = np.nonzero(keep)[0]
        self.o.set_particles(local)
        self.o.sort()
        self.next_flag = self.cur_flag = 0
        # node blocks (regions) in the two block planes at each cut
        ids = self.o.region_ids()
        bx = np.zeros(len(ids), np.int64)
        out = (ctypes.c_uint32 * 3)()
        for i, rid in enumerate(ids):
            lib().oracle_unpack(int(rid), out)
            bx[i] = out[self.axis] >> 2
        for side, cut in ((0, self.lo), (1, self.hi)):
            self.halo_ids[side] = ids[(bx == cut - 1) | (bx == cut)] if cut not in (INT32_MIN, INT32_MAX) else ids[:0]

    def emigrant_count(self, side):
        return len(self.emig[side])

    def emigrants_pack(self, side, buf):
        idx = self.emig[side]
        rec = np.zeros(len(idx), REC)
        for k in ParticleComponents.__slots__:
            rec[k] = getattr(self.comps, k)[idx]
        rec["gid"] = self.gids[idx]
        buf[: rec.nbytes] = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
        self.emig[side] = idx[:0]
        self._drop.append(idx)
        if all(len(e) == 0 for e in self.emig):
            drop = np.concatenate(self._drop)
            keep = np.ones(len(self.gids), bool)
            keep[drop] = False
            for k in ParticleComponents.__slots__:
                setattr(self.comps, k, getattr(self.comps, k)[keep])
            self.gids = self.gids[keep]
            self._drop = []

    def immigrants_add(self, buf, n):
        rec = np.frombuffer(buf[: n * REC.itemsize].numpy().tobytes(), dtype=REC)
        for k in ParticleComponents.__slots__:
            setattr(self.comps, k, np.concatenate([getattr(self.comps, k), rec[k]]))
        self.gids = np.concatenate([self.gids, rec["gid"]])

    def halo_count(self, side):
        return len(self.halo_ids[side])

    def dt_bound_export(self, buf):
        bound = np.float32(self.o.estimate_dt(float(np.finfo(np.float32).max))) if len(self.local_idx) else np.finfo(np.float32).max
        buf[:4] = torch.from_numpy(np.array([bound], np.float32).view(np.uint8).copy())

    def flags_export(self, buf):
        buf[:4] = torch.from_numpy(np.array([self.next_flag], np.int32).view(np.uint8).copy())

    def flags_import(self, buf):
        self.cur_flag = int(buf[:4].numpy().view(np.int32)[0])

    def substep_begin(self, params, gravity, dt):
        self.params, self.g, self.dt = params, gravity, dt
        if self.cur_flag:
            return
        self.o.sort()
        self.o.p2g(dt, gravity)

    def halo_pack(self, side, buf):
        if self.cur_flag:
            return
        rec = np.zeros(len(self.halo_ids[side]), HALO)
        for i, rid in enumerate(self.halo_ids[side]):
            rec["key"][i] = rid
            rec["nodes"][i] = self.o.get_block_nodes(rid)
        buf[: rec.nbytes] = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())

    def halo_merge(self, side, buf, n):
        if self.cur_flag:
            return
        rec = np.frombuffer(buf[: n * HALO.itemsize].numpy().tobytes(), dtype=HALO)
        for r in rec:
            self.o.add_block_nodes(r["key"], r["nodes"])

    def substep_finish(self, fuse_next=False):
        if self.cur_flag:
            self.next_flag = 1
            return
        self.o.refresh_velocities(self.dt, self.g)
        self.o.grid_update(self.dt, self.params.boundary_handling, self.params.simulation_dofs)
        self.o.g2p(self.dt)
        got = self.o.get_particles()
        for k in ParticleComponents.__slots__:
            getattr(self.comps, k)[self.local_idx] = getattr(got, k)
        self.steps_done += 1
        b = self._block_index()
        self.next_flag = int(((b < self.lo) | (b >= self.hi)).any())

    def status(self):
        return self.steps_done, self.next_flag


def _scene():
    sc = scenes.elastic_cube(10)
    sc.particles.velocity["vector"] *= 25.0   # fast spin: particles cross the cut within a few substeps
    return sc


def _worker(rank, world, port, nsteps, dt, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = _scene()
    h = sc.cell_width
    cell = np.round(sc.particles.position["point"][:, 0] / np.float32(h)).astype(np.int64) - 1
    cuts = block_cuts_balanced(cell >> 2, world)
    mine = ((cell >> 2) >= cuts[rank]) & ((cell >> 2) < cuts[rank + 1])
    comps = ParticleComponents(0)
    for k in ParticleComponents.__slots__:
        setattr(comps, k, np.ascontiguousarray(getattr(sc.particles, k)[mine]))
    backend = OracleSlabBackend(sc, comps, np.nonzero(mine)[0], 0, cuts[rank], cuts[rank + 1])
    st = SlabStepper(backend, rank, world, chunk=4)
    dts = []
    if nsteps > 0:
        st.substeps(sc.params, sc.gravity, dt, nsteps)
    else:   # adaptive: -nsteps frames of params.dt
        for _ in range(-nsteps):
            dts += st.step(sc.params, sc.gravity)[1]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gid=backend.gids, pos=backend.comps.position["point"], vel=backend.comps.velocity["vector"],
             F=backend.comps.volume["deformation_gradient"], rebins=st.num_rebins, n0=int(mine.sum()), dts=np.array(dts))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_slabs_match_single_domain(tmp_path):
    from oracle import Oracle
    nsteps, dt, world = 24, 2.0e-4, 2
    mp.spawn(_worker, args=(world, _free_port(), nsteps, dt, str(tmp_path)), nprocs=world, join=True)
    sc = _scene()
    ref = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes)
    ref.load_scene(sc)
    for _ in range(nsteps):
        ref.substep_fixed(sc.params, sc.gravity, dt)
    r = ref.get_particles()
    parts = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    gid = np.concatenate([p["gid"] for p in parts]).astype(np.int64)
    assert sorted(gid.tolist()) == list(range(len(sc.particles))), "every particle owned by exactly one rank"
    pos = np.concatenate([p["pos"] for p in parts])[np.argsort(gid)]
    vel = np.concatenate([p["vel"] for p in parts])[np.argsort(gid)]
    F = np.concatenate([p["F"] for p in parts])[np.argsort(gid)]
    vmax = np.abs(r.velocity["vector"]).max()
    assert np.abs(pos - r.position["point"]).max() / sc.cell_width < 2e-5
    assert np.abs(vel - r.velocity["vector"]).max() / vmax < 2e-5
    assert np.abs(F - r.volume["deformation_gradient"]).max() < 2e-5
    # the cloud rotates through the cut: particles must have migrated, i.e. collective re-bins happened
    assert max(int(p["rebins"]) for p in parts) >= 2
    assert any(len(p["gid"]) != int(p["n0"]) for p in parts)


def test_two_slabs_adaptive_step_matches_single_domain(tmp_path):
    """`SlabStepper.step`: the dt bound is the MIN over both ranks, so the substep lengths are the single-domain ones."""
    from oracle import Oracle
    frames, world = 2, 2
    mp.spawn(_worker, args=(world, _free_port(), -frames, 0.0, str(tmp_path)), nprocs=world, join=True)
    sc = _scene()
    ref = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes)
    ref.load_scene(sc)
    want_dts = []
    for _ in range(frames):
        ref.step(sc.params, sc.gravity)
        want_dts += ref.last_dts().tolist()
    r = ref.get_particles()
    parts = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    for p in parts:
        assert len(p["dts"]) == len(want_dts) and np.allclose(p["dts"], want_dts, rtol=1e-6), (p["dts"], want_dts)
    gid = np.concatenate([p["gid"] for p in parts]).astype(np.int64)
    assert sorted(gid.tolist()) == list(range(len(sc.particles)))
    pos = np.concatenate([p["pos"] for p in parts])[np.argsort(gid)]
    vel = np.concatenate([p["vel"] for p in parts])[np.argsort(gid)]
    vmax = np.abs(r.velocity["vector"]).max()
    assert np.abs(pos - r.position["point"]).max() / sc.cell_width < 5e-5
    assert np.abs(vel - r.velocity["vector"]).max() / vmax < 5e-5


def test_cut_helpers():
    assert block_cuts_even(1, 201, 8) == [INT32_MIN, 26, 51, 76, 101, 126, 151, 176, INT32_MAX]
    cuts = block_cuts_balanced(np.repeat(np.arange(10), 100), 2)
    assert cuts[0] == INT32_MIN and cuts[-1] == INT32_MAX and cuts[1] == 5
