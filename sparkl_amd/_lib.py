"""Loader of libsparkl_hip.so (the HIP product library).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPARKL_HIP_LIB") or os.path.join(_HERE, "lib", "libsparkl_hip.so")  # env override: A/B builds of the same HIP library
_LIB = None

# every symbol include/sparkl_hip.h declares (tests/test_abi_layout.py cross-checks against the header text)
SYMBOLS = [
    "sparkl_hip_pipeline_create", "sparkl_hip_pipeline_destroy", "sparkl_hip_particles_upload", "sparkl_hip_particles_append",
    "sparkl_hip_particles_remove_range", "sparkl_hip_particles_truncate", "sparkl_hip_particles_len",
    "sparkl_hip_particles_read_positions", "sparkl_hip_particles_read_velocities", "sparkl_hip_particles_read_volumes",
    "sparkl_hip_particles_read_statuses", "sparkl_hip_particles_read_phases", "sparkl_hip_particles_read_data",
    "sparkl_hip_particles_read_velocity_gradients", "sparkl_hip_particles_set_fracture", "sparkl_hip_particles_read_fracture", "sparkl_hip_particles_read_sorted_ids", "sparkl_hip_host_alloc", "sparkl_hip_host_free",
    "sparkl_hip_readback_begin", "sparkl_hip_readback_wait", "sparkl_hip_reorder_policy_set", "sparkl_hip_reorder_policy_stats", "sparkl_hip_pipeline_step_stats", "sparkl_hip_grid_check", "sparkl_hip_models_set",
    "sparkl_hip_colliders_set_cuboids", "sparkl_hip_colliders_set_heightfields", "sparkl_hip_colliders_set_trimeshes", "sparkl_hip_colliders_set_cuboid_options", "sparkl_hip_colliders_set_heightfield_options",
    "sparkl_hip_colliders_set_trimesh_options", "sparkl_hip_colliders_set_polylines", "sparkl_hip_colliders_set_polyline_options", "sparkl_hip_colliders_enable_penalty_force", "sparkl_hip_pipeline_step_substep_timings", "sparkl_hip_pipeline_step", "sparkl_hip_pipeline_substeps", "sparkl_hip_pipeline_sync",
    "sparkl_hip_pipeline_clock_enable", "sparkl_hip_pipeline_clock_read", "sparkl_hip_grid_num_active_blocks",
    "sparkl_hip_grid_num_particle_blocks", "sparkl_hip_grid_debug_capture", "sparkl_hip_grid_read_nodes",
    "sparkl_hip_last_error_string", "sparkl_hip_version",
    "sparkl_hip_slab_set", "sparkl_hip_halo_record_bytes", "sparkl_hip_migration_record_bytes", "sparkl_hip_rebin",
    "sparkl_hip_slab_emigrant_count", "sparkl_hip_slab_emigrants_pack", "sparkl_hip_slab_immigrants_add", "sparkl_hip_dt_bound_export", "sparkl_hip_flags_export",
    "sparkl_hip_flags_import", "sparkl_hip_substep_begin", "sparkl_hip_halo_count", "sparkl_hip_halo_pack", "sparkl_hip_halo_merge",
    "sparkl_hip_substep_finish", "sparkl_hip_substep_finish_fused", "sparkl_hip_substep_status",
    "sparkl_hip_comm_unique_id", "sparkl_hip_comm_init", "sparkl_hip_comm_init_all", "sparkl_hip_particles_has_fracture", "sparkl_hip_overlap_info", "sparkl_hip_comm_allreduce_sum", "sparkl_hip_group_substeps", "sparkl_hip_group_step", "sparkl_hip_group_rebins", "sparkl_hip_pipeline_stream", "sparkl_hip_slab_balance", "sparkl_hip_slab_get",
]


def build(force=False, extra=""):
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("sparkl_hip.hip", "kernels.h", "device_math.h", "probes.h", "probe_hooks.h", "host_transport.h", "Makefile")] + [os.path.join(_HERE, "..", "include", "sparkl_hip.h")]
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", src_dir, "-s"] + (["-B"] if force else []) + ([f"EXTRA={extra}"] if extra else [])
        subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). sparkl_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            if not os.environ.get("SPARKL_HIP_LIB_AB"):   # A/B runs against libraries of older revisions (tools/build_rev.sh)
                getattr(L, s)  # AttributeError if the library does not export it
        L.sparkl_hip_last_error_string.restype = C.c_char_p
        L.sparkl_hip_version.restype = C.c_char_p
        L.sparkl_hip_particles_len.restype = C.c_uint64
        L.sparkl_hip_grid_num_active_blocks.restype = C.c_uint64
        L.sparkl_hip_grid_num_particle_blocks.restype = C.c_uint64
        L.sparkl_hip_particles_len.argtypes = [C.c_void_p]
        L.sparkl_hip_grid_num_active_blocks.argtypes = [C.c_void_p]
        L.sparkl_hip_grid_num_particle_blocks.argtypes = [C.c_void_p]
        L.sparkl_hip_halo_record_bytes.restype = C.c_uint64
        if hasattr(L, "sparkl_hip_host_free"):
            L.sparkl_hip_host_free.argtypes = [C.c_void_p]
        L.sparkl_hip_migration_record_bytes.restype = C.c_uint64
        if hasattr(L, "sparkl_hip_group_rebins"):
            L.sparkl_hip_group_rebins.restype = C.c_uint64
            L.sparkl_hip_group_rebins.argtypes = [C.c_void_p]
            L.sparkl_hip_pipeline_stream.restype = C.c_void_p
            L.sparkl_hip_pipeline_stream.argtypes = [C.c_void_p]
        L.sparkl_hip_pipeline_destroy.argtypes = [C.c_void_p]
        L.sparkl_hip_pipeline_destroy.restype = None
        _LIB = L
    return _LIB


class SparklHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sparkl_hip error {code}: {msg}")
        self.code = code


def check(code):
    if code != 0:
        raise SparklHipError(code, lib().sparkl_hip_last_error_string().decode("utf-8", "replace"))
