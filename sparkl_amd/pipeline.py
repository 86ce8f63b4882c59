"""Host-side mirror of the reference pipeline interface, bound to libsparkl_hip.so through ctypes.

  MpmPipeline.step(params, gravity, ...)   <->  MpmPipeline::step      src/pipelines/mpm_pipeline.rs:15-42
                                                CudaMpmPipeline::step  src/cuda/cuda_mpm_pipeline.rs:262-271
  HipParticleSet                           <->  CudaParticleSet        src/cuda/cuda_particle_set.rs:12-177
  models / colliders upload                <->  CudaParticleModelSet::from_model_set, CudaColliderSet::from_collider_set

The compute is entirely inside the HIP library; this file only marshals arguments.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import check, lib
from .particles import ParticleComponents


class _Pinned:
    def __init__(self, L, ptr):
        self.L, self.ptr = L, ptr

    def __del__(self):
        try:
            self.L.sparkl_hip_host_free(self.ptr)
        except Exception:
            pass


class MpmPipeline:
    """One GPU's MPM pipeline (CudaMpmPipeline + SingleGpuMpmContext rolled into an opaque handle)."""

    def __init__(self, cell_width, device_id=0, domain_log2_bytes=30, stream=None):
        self.L = lib()
        desc = abi.PipelineDescC(device_id, cell_width, domain_log2_bytes, stream)
        h = C.c_void_p()
        check(self.L.sparkl_hip_pipeline_create(C.byref(desc), C.byref(h)))
        self.h = h
        self.cell_width = cell_width
        self.stream = stream   # hipStream_t handle the kernels run on (None: a library-owned non-blocking stream)

    def close(self):
        if getattr(self, "h", None):
            self.L.sparkl_hip_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- multi-GPU slabs (see sparkl_amd/slab.py)
    def set_slab(self, axis, block_lo, block_hi, id_offset=0):
        """Own the home blocks [block_lo, block_hi) along `axis`; call before upload_particles.
        In slab mode the read_* methods return device order; read_sorted_particle_ids() gives the global ids."""
        from .slab import SlabC
        s = SlabC(axis, block_lo, block_hi, 0, id_offset)
        check(self.L.sparkl_hip_slab_set(self.h, C.byref(s)))

    # ---- the exchange inside the library (RCCL): see sparkl_hip_group_substeps in the header
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (ncclGetUniqueId): create on one rank, hand to every rank's comm_init."""
        buf = (C.c_uint8 * 128)()
        check(lib().sparkl_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(self.L.sparkl_hip_comm_init(self.h, buf, C.c_int(rank), C.c_int(world)))

    @staticmethod
    def comm_init_all(members, unique_id, first_rank, world):
        """One host thread, one member per device: member i joins as rank first_rank + i (grouped ncclCommInitRank), the shape of
        `CudaMpmPipeline::step(.., &mut [SingleGpuMpmContext])` (cuda_mpm_pipeline.rs:262-271)."""
        arr = (C.c_void_p * len(members))(*[m.h for m in members])
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(members[0].L.sparkl_hip_comm_init_all(arr, C.c_uint32(len(members)), buf, C.c_int(first_rank), C.c_int(world)))

    def has_fracture(self):
        """True once a `ParticleFracture` component was uploaded (set_fracture): part of the state a checkpoint must carry."""
        return bool(self.L.sparkl_hip_particles_has_fracture(self.h))

    def overlap_info(self):
        """(chosen, in_line_us, overlapped_us): the group loop's schedule between RCCL ranks (1 overlapped, 0 in line, -1 undecided) and the
        per-substep wall clock of the two calibration batches (MAX over the ranks)."""
        ch, us = C.c_int(-1), (C.c_double * 2)()
        check(self.L.sparkl_hip_overlap_info(self.h, C.byref(ch), us))
        return int(ch.value), float(us[0]), float(us[1])

    def comm_allreduce_sum(self, value):
        v = C.c_longlong(int(value))
        check(self.L.sparkl_hip_comm_allreduce_sum(self.h, C.byref(v)))
        return int(v.value)

    @staticmethod
    def group_substeps(members, params, gravity, dt, n):
        """n fixed-length substeps of the slab pipelines `members` (ascending along the slab axis) in lockstep, halo exchange,
        re-bin agreement and particle migration inside the library (RCCL between ranks, device copies inside a process)."""
        arr = (C.c_void_p * len(members))(*[m.h for m in members])
        g = (C.c_float * 3)(*[float(x) for x in gravity])
        check(members[0].L.sparkl_hip_group_substeps(arr, C.c_uint32(len(members)), C.byref(params), g, C.c_float(dt), C.c_uint32(n)))

    @staticmethod
    def group_step(members, params, gravity):
        """`MpmPipeline::step` across the slab pipelines `members`: returns (simulated time, number of substeps)."""
        arr = (C.c_void_p * len(members))(*[m.h for m in members])
        g = (C.c_float * 3)(*[float(x) for x in gravity])
        sim, nsub = C.c_float(0.0), C.c_uint32(0)
        check(members[0].L.sparkl_hip_group_step(arr, C.c_uint32(len(members)), C.byref(params), g, C.byref(sim), C.byref(nsub)))
        return sim.value, int(nsub.value)

    def direct_info(self):
        """Peer-direct halo exchange of the group loop (kernels.h, PeerBox): {"ready", "epochs", "failed", "members", "arena_fine_grained", "handshakes", "chosen", "timed_us_per_substep"}."""
        out = (C.c_ulonglong * 4)()
        self.L.sparkl_hip_debug_direct(self.h, out)
        choice, us = C.c_int(-1), (C.c_double * 3)()
        self.L.sparkl_hip_debug_direct_choice(self.h, C.byref(choice), us)
        # "chosen": what the three-way timing of the job kept (None: nothing timed -- inside one process, or forced), "timed_us_per_substep": the MAX over
        # the ranks of one undisturbed 16-substep batch on the direct exchange, the staged exchange in line and the staged exchange overlapped
        return {"ready": bool(out[0]), "epochs": int(out[1]), "failed": bool(out[2] & 1), "members": int(out[3]), "arena_fine_grained": bool(out[2] & 2), "handshakes": int(out[2] >> 8),
                "chosen": {-1: None, 0: "staged", 1: "direct"}[int(choice.value)], "timed_us_per_substep": {"direct": us[0], "staged_in_line": us[1], "staged_overlapped": us[2]}}

    def slab_balance(self, enable=True):
        check(self.L.sparkl_hip_slab_balance(self.h, int(bool(enable))))

    def slab_get(self):
        """(axis, block_lo, block_hi, number of times a cut of this slab was moved by the re-balancing)."""
        from .slab import SlabC
        s, moves = SlabC(0, 0, 0, 0, 0), C.c_uint64(0)
        check(self.L.sparkl_hip_slab_get(self.h, C.byref(s), C.byref(moves)))
        return int(s.axis), int(s.block_lo), int(s.block_hi), int(moves.value)

    def group_rebins(self):
        return int(self.L.sparkl_hip_group_rebins(self.h))

    # ---- CudaParticleModelSet / CudaColliderSet
    def set_models(self, model_set):
        arr = model_set.to_c_array()
        check(self.L.sparkl_hip_models_set(self.h, C.c_uint32(len(model_set)), arr))
        self._uploaded_models = bytes(arr)[: len(model_set) * C.sizeof(abi.ParticleModelC)]

    def ensure_models_are_uploaded(self, model_set):
        """`CudaParticleModelSet::ensure_models_are_uploaded` (cuda_particle_model_set.rs:61-84): the user may have edited a
        model between two steps; re-upload only when the core models differ from what the device holds.  Returns True if it did."""
        now = bytes(model_set.to_c_array())[: len(model_set) * C.sizeof(abi.ParticleModelC)]
        if now == getattr(self, "_uploaded_models", None):
            return False
        self.set_models(model_set)
        return True

    def set_colliders(self, cuboids):
        arr = (abi.CuboidColliderC * max(len(cuboids), 1))(*cuboids)
        check(self.L.sparkl_hip_colliders_set_cuboids(self.h, C.c_uint32(len(cuboids)), arr))

    def set_heightfields(self, heightfields):
        """`CudaColliderSet::from_collider_set` for heightfield colliders (abi.heightfield(..)); applied after the cuboids."""
        arr = (abi.HeightFieldColliderC * max(len(heightfields), 1))(*heightfields)
        check(self.L.sparkl_hip_colliders_set_heightfields(self.h, C.c_uint32(len(heightfields)), arr))

    def set_trimeshes(self, trimeshes):
        """`CudaColliderSet::from_collider_set` for triangle meshes (abi.trimesh(..)); applied after the heightfields."""
        arr = (abi.TriMeshColliderC * max(len(trimeshes), 1))(*trimeshes)
        check(self.L.sparkl_hip_colliders_set_trimeshes(self.h, C.c_uint32(len(trimeshes)), arr))

    def set_polylines(self, polylines):
        """`CudaColliderSet::from_collider_set` for polyline colliders (abi.polyline(..)); applied after the triangle meshes."""
        arr = (abi.PolylineColliderC * max(len(polylines), 1))(*polylines)
        check(self.L.sparkl_hip_colliders_set_polylines(self.h, C.c_uint32(len(polylines)), arr))

    def set_collider_options(self, kind, options):
        """`CudaColliderSet::from_collider_set(.., options)`: one abi.collider_options(..) per collider of `kind` ("cuboid", "heightfield",
        "trimesh", "polyline"), in the order of the matching set_* call (which resets the options of its kind to the defaults)."""
        fn = {"cuboid": self.L.sparkl_hip_colliders_set_cuboid_options, "heightfield": self.L.sparkl_hip_colliders_set_heightfield_options,
              "trimesh": self.L.sparkl_hip_colliders_set_trimesh_options, "polyline": self.L.sparkl_hip_colliders_set_polyline_options}[kind]
        arr = (abi.ColliderOptionsC * max(len(options), 1))(*options)
        check(fn(self.h, C.c_uint32(len(options)), arr))

    def enable_penalty_force(self, on=True):
        """`penalty_stiffness` acts only behind this switch: the reference has the force under `if false` (particle_updater.rs:203-221)."""
        check(self.L.sparkl_hip_colliders_enable_penalty_force(self.h, C.c_int(1 if on else 0)))

    # ---- CudaParticleSet
    def _args(self, c):
        return (C.c_uint64(len(c)), abi.ptr(c.status, abi.STATUS_DTYPE), abi.ptr(c.position, abi.POSITION_DTYPE),
                abi.ptr(c.velocity, abi.VELOCITY_DTYPE), abi.ptr(c.volume, abi.VOLUME_DTYPE), abi.ptr(c.phase, abi.PHASE_DTYPE),
                abi.ptr(c.data, abi.DATA_DTYPE), abi.ptr(c.velocity_gradient, abi.VELGRAD_DTYPE))

    def upload_particles(self, comps):
        """`CudaParticleSet::from_particles` on `extract_particles_components(..)` output."""
        check(self.L.sparkl_hip_particles_upload(self.h, *self._args(comps)))

    def append_particles(self, comps):
        check(self.L.sparkl_hip_particles_append(self.h, *self._args(comps)))

    def remove_range(self, start, end):
        check(self.L.sparkl_hip_particles_remove_range(self.h, C.c_uint64(start), C.c_uint64(end)))

    def truncate(self, new_len):
        check(self.L.sparkl_hip_particles_truncate(self.h, C.c_uint64(new_len)))

    def __len__(self):
        return int(self.L.sparkl_hip_particles_len(self.h))

    def _read(self, fn, dtype):
        n = len(self)
        out = np.zeros(n, dtype)
        check(fn(self.h, out.ctypes.data_as(C.c_void_p), C.c_uint64(n)))
        return out

    def read_positions(self):
        return self._read(self.L.sparkl_hip_particles_read_positions, abi.POSITION_DTYPE)

    def read_velocities(self):
        return self._read(self.L.sparkl_hip_particles_read_velocities, abi.VELOCITY_DTYPE)

    def read_volumes(self):
        return self._read(self.L.sparkl_hip_particles_read_volumes, abi.VOLUME_DTYPE)

    def read_statuses(self):
        return self._read(self.L.sparkl_hip_particles_read_statuses, abi.STATUS_DTYPE)

    def read_phases(self):
        return self._read(self.L.sparkl_hip_particles_read_phases, abi.PHASE_DTYPE)

    def read_data(self):
        return self._read(self.L.sparkl_hip_particles_read_data, abi.DATA_DTYPE)

    def set_fracture(self, fracture):
        """`ParticleFracture` of the resident particles (caller order): crack_propagation_factor, crack_threshold.  Only read under
        DamageModel::ModifiedEigenerosion (grid_to_particle.rs:65-77)."""
        fracture = np.ascontiguousarray(fracture, dtype=abi.FRACTURE_DTYPE)
        check(self.L.sparkl_hip_particles_set_fracture(self.h, abi.ptr(fracture, abi.FRACTURE_DTYPE), C.c_uint64(len(fracture))))

    def read_fracture(self):
        return self._read(self.L.sparkl_hip_particles_read_fracture, abi.FRACTURE_DTYPE)

    def read_velocity_gradients(self):
        return self._read(self.L.sparkl_hip_particles_read_velocity_gradients, abi.VELGRAD_DTYPE)

    def read_sorted_particle_ids(self):
        return self._read(self.L.sparkl_hip_particles_read_sorted_ids, np.dtype("<u4"))

    def reorder_policy_set(self, reorder_limit=-1.0, rebin_limit=0.0):
        """Limits in excess run heads: 0 = off, < 0 = built-in default (see sparkl_hip_reorder_policy_set)."""
        check(self.L.sparkl_hip_reorder_policy_set(self.h, C.c_float(reorder_limit), C.c_float(rebin_limit)))

    def step_substep_timings(self):
        """`timings.substeps` of the last step(.., timings=True): a list of abi.SubstepTimingsC, one per substep."""
        n = C.c_uint32(0)
        check(self.L.sparkl_hip_pipeline_step_substep_timings(self.h, None, C.c_uint32(0), C.byref(n)))
        arr = (abi.SubstepTimingsC * max(n.value, 1))()
        check(self.L.sparkl_hip_pipeline_step_substep_timings(self.h, arr, C.c_uint32(n.value), C.byref(n)))
        return list(arr)[:n.value]

    def step_stats(self):
        """Counters of the adaptive step: fused-scatter hits / misses, look-ahead substeps / mismatches."""
        out = (C.c_uint64 * 7)()
        check(self.L.sparkl_hip_pipeline_step_stats(self.h, out))
        return {"fused_hits": out[0], "fused_misses": out[1], "lookahead_substeps": out[2], "lookahead_mismatches": out[3], "c_regathers": out[4],
                "fluid_fused_substeps": out[5], "fluid_fused_mismatches": out[6]}

    def grid_check(self):
        """The reference's `check_active_blocks` (cuda_sparse_grid.rs:492-650) on the current binning: a dict of violation counts."""
        k = abi.GridCheckC()
        check(self.L.sparkl_hip_grid_check(self.h, C.byref(k)))
        return {name: int(getattr(k, name)) for name, _ in abi.GridCheckC._fields_}

    def reorder_policy_stats(self):
        st = abi.ReorderStatsC()
        check(self.L.sparkl_hip_reorder_policy_stats(self.h, C.byref(st)))
        return {"last_heads": st.last_heads, "base_heads": st.base_heads, "reorders": st.reorders, "policy_rebins": st.policy_rebins}

    # ---- per-frame render readback (testbed_plugin.rs:533-588), overlapped with the next step
    def host_alloc(self, n, dtype):
        """`LockedBuffer::uninitialized(n)`: a page-locked numpy array of n records (freed with the pipeline)."""
        dtype = np.dtype(dtype)
        ptr = C.c_void_p()
        check(self.L.sparkl_hip_host_alloc(C.byref(ptr), C.c_uint64(max(n * dtype.itemsize, 1))))
        buf = (C.c_uint8 * max(n * dtype.itemsize, 1)).from_address(ptr.value)
        buf._owner = _Pinned(self.L, ptr)           # freed when the last numpy view goes away
        return np.frombuffer(buf, dtype=dtype, count=n)

    def readback_begin(self, pos=None, vel=None, phase=None):
        """Snapshot positions / velocities / phases (caller order) into the given arrays; returns at once."""
        def ptr(a, dt):
            if a is None:
                return None
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"] and len(a) == len(self)
            return C.c_void_p(a.ctypes.data)
        self._rb_keep = (pos, vel, phase)
        check(self.L.sparkl_hip_readback_begin(self.h, ptr(pos, abi.POSITION_DTYPE), ptr(vel, abi.VELOCITY_DTYPE), ptr(phase, abi.PHASE_DTYPE),
                                               C.c_uint64(len(self))))

    def readback_wait(self):
        check(self.L.sparkl_hip_readback_wait(self.h))

    def read_particles(self):
        c = ParticleComponents(0)
        c.status, c.position, c.velocity = self.read_statuses(), self.read_positions(), self.read_velocities()
        c.volume, c.phase, c.data = self.read_volumes(), self.read_phases(), self.read_data()
        c.velocity_gradient = self.read_velocity_gradients()
        return c

    def load_scene(self, scene):
        self.set_models(scene.models)
        self.set_colliders(scene.colliders)
        self.set_heightfields(getattr(scene, "heightfields", []))
        self.set_trimeshes(getattr(scene, "trimeshes", []))
        self.set_polylines(getattr(scene, "polylines", []))
        for kind, opts in getattr(scene, "collider_options", {}).items():
            self.set_collider_options(kind, opts)
        self.upload_particles(scene.particles)

    # ---- stepping
    def step(self, params, gravity, timings=False):
        """`MpmPipeline::step` / `CudaMpmPipeline::step`: advance by params.dt, return the simulated time."""
        g = (C.c_float * 3)(*[float(x) for x in gravity])
        t = abi.TimingsC() if timings else None
        sim = C.c_float(0.0)
        check(self.L.sparkl_hip_pipeline_step(self.h, C.byref(params), g, C.byref(t) if timings else None, C.byref(sim)))
        return (sim.value, t) if timings else sim.value

    def substeps(self, params, gravity, dt, n):
        """n fixed-length substeps without host synchronisation inside (extension, see the header)."""
        g = (C.c_float * 3)(*[float(x) for x in gravity])
        check(self.L.sparkl_hip_pipeline_substeps(self.h, C.byref(params), g, C.c_float(dt), C.c_uint32(n)))

    def sync(self):
        check(self.L.sparkl_hip_pipeline_sync(self.h))

    def clock_enable(self, on=True, stride=1):
        """Per-kernel hipEvent spans; stride N > 1 samples every N-th substep."""
        check(self.L.sparkl_hip_pipeline_clock_enable(self.h, (int(stride) if stride > 1 else 1) if on else 0))

    def clock_read(self, reset=True):
        k = abi.KernelClockC()
        check(self.L.sparkl_hip_pipeline_clock_read(self.h, C.byref(k), int(reset)))
        return k

    # ---- sparse grid inspection
    def num_active_blocks(self):
        return int(self.L.sparkl_hip_grid_num_active_blocks(self.h))

    def num_particle_blocks(self):
        return int(self.L.sparkl_hip_grid_num_particle_blocks(self.h))

    def grid_debug_capture(self, on=True):
        check(self.L.sparkl_hip_grid_debug_capture(self.h, int(on)))

    def read_grid_nodes(self):
        n = C.c_uint64(0)
        check(self.L.sparkl_hip_grid_read_nodes(self.h, None, C.c_uint64(0), C.byref(n)))
        out = np.zeros(n.value, abi.GRID_NODE_DTYPE)
        if n.value:
            check(self.L.sparkl_hip_grid_read_nodes(self.h, abi.ptr(out, abi.GRID_NODE_DTYPE), C.c_uint64(n.value), C.byref(n)))
        return out
