"""Slab-decomposed multi-GPU stepping: one process per GPU, torch.distributed (RCCL on ROCm) for the exchange.

The reference's multi-GPU path partitions particles by model index, makes every GPU span the whole domain and
moves whole 4 616-byte halo blocks all-to-all with peer copies (src/cuda/cuda_sparse_grid.rs:255-322,
src/cuda/cuda_mpm_pipeline.rs:407-533; disabled in testbed_plugin.rs:182-189).  Here the domain is cut into slabs
of 4x4x4 node blocks along one axis; a rank only ever talks to its two neighbours:

  per substep   flag all-reduce(max)  ->  P2G  ->  halo pack  ->  send/recv with both neighbours  ->  halo merge
                ->  grid update  ->  G2P
  per re-bin    emigrant counts / payloads to the neighbours, immigrants appended, halo sizes re-negotiated

Nothing here computes physics: the backend (the HIP library, `HipSlabBackend`) does.  The orchestration is written
against a small backend interface so that the CPU test-suite can drive exactly this code with a checker backend
over gloo (tests/test_slab_gloo.py).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check

INT32_MIN, INT32_MAX = -(2 ** 31), 2 ** 31 - 1


class SlabC(C.Structure):
    _fields_ = [("axis", C.c_int32), ("block_lo", C.c_int32), ("block_hi", C.c_int32), ("_pad0", C.c_uint32), ("id_offset", C.c_uint64)]


def block_cuts_even(lo_block, hi_block, world):
    """world+1 cut positions (block indices) splitting [lo_block, hi_block) evenly; the outer cuts are open."""
    edges = [lo_block + (hi_block - lo_block) * r // world for r in range(world + 1)]
    edges[0], edges[-1] = INT32_MIN, INT32_MAX
    return edges


def block_cuts_balanced(block_index_of_particle, world):
    """Cuts on the particle histogram over block planes so every rank owns ~N/world particles."""
    b = np.sort(np.asarray(block_index_of_particle))
    edges = [INT32_MIN]
    for r in range(1, world):
        edges.append(int(b[len(b) * r // world]))
    edges.append(INT32_MAX)
    for r in range(1, world):  # strictly increasing
        edges[r] = max(edges[r], edges[r - 1] + 1 if edges[r - 1] != INT32_MIN else edges[r])
    return edges


class HipSlabBackend:
    """The real thing: thin calls into libsparkl_hip.so on torch's current stream."""

    def __init__(self, pipe, device):
        self.pipe, self.L, self.h, self.device = pipe, pipe.L, pipe.h, device
        # pack / merge kernels and the torch-side send / recv / .cpu() must be ordered on ONE stream: the pipeline has to be
        # created on torch's current stream (a library-owned stream would race with the exchange)
        cur = torch.cuda.current_stream(device).cuda_stream
        if pipe.stream is None or int(pipe.stream) != cur:
            raise ValueError("HipSlabBackend: create the MpmPipeline with stream=torch.cuda.current_stream(device).cuda_stream "
                             f"(pipeline stream {pipe.stream!r}, torch current stream {cur:#x})")
        self.halo_record_bytes = int(self.L.sparkl_hip_halo_record_bytes())
        self.migration_record_bytes = int(self.L.sparkl_hip_migration_record_bytes())

    def new_buffer(self, nbytes):
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)

    def _ptr(self, t):
        return C.c_void_p(t.data_ptr())

    def set_slab(self, axis, lo, hi, id_offset):
        s = SlabC(axis, lo, hi, 0, id_offset)
        check(self.L.sparkl_hip_slab_set(self.h, C.byref(s)))

    def rebin(self):
        check(self.L.sparkl_hip_rebin(self.h))

    def _count(self, fn, side):
        n = C.c_uint64(0)
        check(fn(self.h, side, C.byref(n)))
        return int(n.value)

    def emigrant_count(self, side):
        return self._count(self.L.sparkl_hip_slab_emigrant_count, side)

    def emigrants_pack(self, side, buf):
        check(self.L.sparkl_hip_slab_emigrants_pack(self.h, side, self._ptr(buf)))

    def immigrants_add(self, buf, n):
        check(self.L.sparkl_hip_slab_immigrants_add(self.h, self._ptr(buf), C.c_uint64(n)))

    def halo_count(self, side):
        return self._count(self.L.sparkl_hip_halo_count, side)

    def halo_pack(self, side, buf):
        check(self.L.sparkl_hip_halo_pack(self.h, side, self._ptr(buf)))

    def halo_merge(self, side, buf, n):
        check(self.L.sparkl_hip_halo_merge(self.h, side, self._ptr(buf), C.c_uint64(n)))

    def dt_bound_export(self, buf):
        check(self.L.sparkl_hip_dt_bound_export(self.h, self._ptr(buf)))

    def flags_export(self, buf):
        check(self.L.sparkl_hip_flags_export(self.h, self._ptr(buf)))

    def flags_import(self, buf):
        check(self.L.sparkl_hip_flags_import(self.h, self._ptr(buf)))

    def substep_begin(self, params, gravity, dt):
        g = (C.c_float * 3)(*[float(x) for x in gravity])
        check(self.L.sparkl_hip_substep_begin(self.h, C.byref(params), g, C.c_float(dt)))

    def substep_finish(self, fuse_next=False):
        """fuse_next: G2P also scatters the next substep's tiles (fixed dt); the next substep_begin is then a no-op."""
        check((self.L.sparkl_hip_substep_finish_fused if fuse_next else self.L.sparkl_hip_substep_finish)(self.h))

    def status(self):
        done, need = C.c_uint32(0), C.c_uint32(0)
        check(self.L.sparkl_hip_substep_status(self.h, C.byref(done), C.byref(need)))
        return int(done.value), int(need.value)


class SlabStepper:
    """Drives one rank's backend through collective re-bins and substeps."""

    def __init__(self, backend, rank, world, group=None, chunk=16):
        self.b, self.rank, self.world, self.group, self.chunk = backend, rank, world, group, chunk
        self.nbr = [rank - 1 if rank > 0 else None, rank + 1 if rank < world - 1 else None]  # side 0 = low, 1 = high
        self.staged = world > 1 and dist.get_backend(group) == "gloo"
        self.flag = backend.new_buffer(16)
        self.send = [None, None]
        self.recv = [None, None]
        self.n_send = [0, 0]
        self.n_recv = [0, 0]
        self.binned = False
        self.num_rebins = 0

    # ---- transport (gloo cannot touch device memory: stage through the host; nccl uses the buffers in place)
    def _p2p(self, sends, recvs):
        """sends / recvs: {side: tensor}.  Both neighbours at once, deadlock-free."""
        if self.world == 1:
            return
        ops, post = [], []
        for side, t in sends.items():
            if self.nbr[side] is None:
                continue
            src = t.cpu() if (self.staged and t.is_cuda) else t
            ops.append(dist.P2POp(dist.isend, src, self.nbr[side], group=self.group))
        for side, t in recvs.items():
            if self.nbr[side] is None:
                continue
            if self.staged and t.is_cuda:
                tmp = torch.empty(t.shape, dtype=t.dtype)
                post.append((t, tmp))
                ops.append(dist.P2POp(dist.irecv, tmp, self.nbr[side], group=self.group))
            else:
                ops.append(dist.P2POp(dist.irecv, t, self.nbr[side], group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for dst, tmp in post:
            dst.copy_(tmp)

    def _allreduce_max_i32(self, t_u8):
        """t_u8: >= 4-byte uint8 buffer viewed as one int32."""
        if self.world == 1:
            return
        v = t_u8[:4].view(torch.int32)
        if self.staged and v.is_cuda:
            c = v.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
            v.copy_(c)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)

    def _allreduce_min_f32_bits(self, t_u8):
        """t_u8[:4] holds the bits of a non-negative f32: integer MIN == float MIN.  Returns the float (host sync)."""
        v = t_u8[:4].view(torch.int32)
        if self.world > 1:
            if self.staged and v.is_cuda:
                c = v.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MIN, group=self.group)
                v.copy_(c)
            else:
                dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.group)
        return float(np.array([int(v.cpu().item())], dtype=np.int32).view(np.float32)[0])

    def _allreduce_max_host(self, value):
        """max over ranks of a host integer."""
        if self.world == 1:
            return int(value)
        t = torch.tensor([int(value)], dtype=torch.int64)
        if self.staged or not self.flag.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return int(t.item())
        d = t.to(self.flag.device)
        dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.group)
        return int(d.item())

    def _swap_counts(self, counts):
        """counts[side] -> what each neighbour announced for the matching side."""
        out = [0, 0]
        if self.world == 1:
            return out
        s = {side: torch.tensor([counts[side]], dtype=torch.int64) for side in (0, 1)}
        r = {side: torch.zeros(1, dtype=torch.int64) for side in (0, 1)}
        if not self.staged and self.flag.is_cuda:
            s = {k: v.to(self.flag.device) for k, v in s.items()}
            r = {k: v.to(self.flag.device) for k, v in r.items()}
        self._p2p(s, r)
        for side in (0, 1):
            if self.nbr[side] is not None:
                out[side] = int(r[side].item())
        return out

    # ---- collective re-bin with particle migration
    def rebin(self):
        b = self.b
        while True:
            b.rebin()
            n_out = [b.emigrant_count(0) if self.nbr[0] is not None else 0, b.emigrant_count(1) if self.nbr[1] is not None else 0]
            if self._allreduce_max_host(n_out[0] + n_out[1]) == 0:
                break
            n_in = self._swap_counts(n_out)
            rb = b.migration_record_bytes
            sends = {side: b.new_buffer(n_out[side] * rb) for side in (0, 1)}
            recvs = {side: b.new_buffer(n_in[side] * rb) for side in (0, 1)}
            for side in (0, 1):
                if n_out[side]:
                    b.emigrants_pack(side, sends[side])
            self._p2p({s: sends[s][: max(n_out[s] * rb, 16)] for s in (0, 1) if n_out[s]}, {s: recvs[s][: max(n_in[s] * rb, 16)] for s in (0, 1) if n_in[s]})
            for side in (0, 1):
                if n_in[side]:
                    b.immigrants_add(recvs[side], n_in[side])
        # halo buffers for the new block lists
        hb = b.halo_record_bytes
        self.n_send = [b.halo_count(0) if self.nbr[0] is not None else 0, b.halo_count(1) if self.nbr[1] is not None else 0]
        self.n_recv = self._swap_counts(self.n_send)
        for side in (0, 1):
            self.send[side] = b.new_buffer(self.n_send[side] * hb)
            self.recv[side] = b.new_buffer(self.n_recv[side] * hb)
        self.binned = True
        self.num_rebins += 1

    # ---- substeps
    def _one_substep(self, params, gravity, dt, fuse_next=False):
        b, hb = self.b, self.b.halo_record_bytes
        b.flags_export(self.flag)
        self._allreduce_max_i32(self.flag)
        b.flags_import(self.flag)
        b.substep_begin(params, gravity, dt)
        for side in (0, 1):
            if self.n_send[side]:
                b.halo_pack(side, self.send[side])
        self._p2p({s: self.send[s][: self.n_send[s] * hb] for s in (0, 1) if self.n_send[s]},
                  {s: self.recv[s][: self.n_recv[s] * hb] for s in (0, 1) if self.n_recv[s]})
        for side in (0, 1):
            if self.n_recv[side]:
                b.halo_merge(side, self.recv[side], self.n_recv[side])
        b.substep_finish(fuse_next)

    def step(self, params, gravity):
        """`MpmPipeline::step` across the slabs: advances by params.dt in adaptive substeps whose length follows the rule of
        mls_solver.rs:33-59 with the dt bound taken over ALL ranks' particles (timestep_estimator.rs:10-51).  Returns
        (simulated time, list of substep lengths); every rank gets the same values."""
        f32 = np.float32
        if not self.binned:
            self.rebin()
        total = f32(params.dt)
        if total == 0:
            return float(total), []
        min_dt = total / f32(params.max_num_substeps)
        remaining, dts = total, []
        while remaining > 0:
            self.b.dt_bound_export(self.flag)
            bound = f32(self._allreduce_min_f32_bits(self.flag))
            dt = min(min(remaining, f32(params.max_substep_dt)), bound)
            if dt < min_dt and remaining > min_dt:
                dt = min_dt
            self.substeps(params, gravity, float(dt), 1)
            dts.append(float(dt))
            remaining = f32(remaining - dt)
            if params.stop_after_one_substep:
                return float(dt), dts
        return float(total), dts

    def substeps(self, params, gravity, dt, n):
        """n fixed-length substeps; re-bins (collectively) whenever any rank's particles leave their tile margin."""
        if not self.binned:
            self.rebin()
        base, _ = self.b.status()
        done = 0
        guard = 0
        while done < n:
            batch = min(self.chunk, n - done)
            for j in range(batch):
                # all substeps of a batch share dt/params, so G2P can scatter the next substep's tiles right away
                self._one_substep(params, gravity, dt, fuse_next=(j + 1 < batch))
            now, need = self.b.status()
            ran = now - base
            # the flag is global, so every rank ran the same substeps
            need_any = self._allreduce_max_host(1 if (ran < done + batch or need) else 0)
            done = ran
            if need_any:
                self.rebin()
                guard += 1
                if guard > 2 * n + 16:
                    raise RuntimeError("slab re-bin loop made no progress")
