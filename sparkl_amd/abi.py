"""ctypes / numpy mirrors of include/sparkl_hip.h.

Every dtype here is byte-identical to the `#[repr(C)]` Rust struct cited in the header
(reference: src_core/dynamics/particle.rs:18-219, src_core/dynamics/solver/solver_parameters.rs:42-54,
src_core/dynamics/models/*.rs).  tests/test_abi_layout.py checks the sizes against the header.
"""
import ctypes as C

import numpy as np

# ----------------------------------------------------------------------------- particle components
POSITION_DTYPE = np.dtype([("point", "<f4", (3,))])                      # 12 B
VELOCITY_DTYPE = np.dtype([("vector", "<f4", (3,))])                     # 12 B
VOLUME_DTYPE = np.dtype([("mass", "<f4"), ("volume0", "<f4"), ("radius0", "<f4"),
                         ("deformation_gradient", "<f4", (9,)),
                         ("plastic_deformation_gradient_det", "<f4")])   # 52 B
STATUS_DTYPE = np.dtype([("failed", "u1"), ("is_static", "u1"), ("kinematic_vel_enabled", "u1"),
                         ("_pad0", "u1"), ("kinematic_vel", "<f4", (3,)), ("model_index", "<u8")])  # 24 B
PHASE_DTYPE = np.dtype([("phase", "<f4"), ("psi_pos", "<f4")])           # 8 B
FRACTURE_DTYPE = np.dtype([("crack_propagation_factor", "<f4"), ("crack_threshold", "<f4")])   # 8 B, optional component
DATA_DTYPE = np.dtype([("nacc_alpha", "<f4"), ("plastic_hardening", "<f4"),
                       ("elastic_hardening", "<f4"), ("log_vol_gain", "<f4")])  # 16 B
VELGRAD_DTYPE = np.dtype([("m", "<f4", (9,))])                           # 36 B
GRID_NODE_DTYPE = np.dtype([("cell", "<i4", (3,)), ("mass", "<f4"), ("momentum", "<f4", (3,)),
                            ("velocity", "<f4", (3,))])                  # 40 B

# BoundaryHandling / DamageModel / SimulationDofs (solver_parameters.rs:7-36)
BOUNDARY_STICK, BOUNDARY_FRICTION, BOUNDARY_FRICTION_Z_UP, BOUNDARY_NONE = 0, 1, 2, 3
DAMAGE_CDMPM, DAMAGE_EIGENEROSION, DAMAGE_MODIFIED_EIGENEROSION, DAMAGE_NONE = 0, 1, 2, 3
LOCK_NONE, LOCK_X, LOCK_Y, LOCK_Z = 0, 1, 2, 4

F32_MAX = float(np.finfo(np.float32).max)


class SolverParameters(C.Structure):
    """`SolverParameters` (solver_parameters.rs:42-68); defaults as `impl Default`."""
    _fields_ = [("dt", C.c_float), ("max_substep_dt", C.c_float), ("max_num_substeps", C.c_uint32),
                ("boundary_handling", C.c_int32), ("damage_model", C.c_int32),
                ("force_fluids_volume_recomputation", C.c_uint8),
                ("enable_boundary_particle_projection", C.c_uint8),
                ("stop_after_one_substep", C.c_uint8), ("_pad0", C.c_uint8),
                ("simulation_dofs", C.c_uint32)]

    def __init__(self, **kw):
        super().__init__()
        self.dt = 1.0 / 60.0
        self.max_substep_dt = F32_MAX
        self.max_num_substeps = 1000
        self.boundary_handling = BOUNDARY_FRICTION
        self.damage_model = DAMAGE_NONE
        self.simulation_dofs = LOCK_NONE
        for k, v in kw.items():
            setattr(self, k, v)


# ------------------------------------------------------------------------------------------ models
class CorotatedLinearElasticityC(C.Structure):
    _fields_ = [("split_stress_on_failure", C.c_uint8), ("_pad", C.c_uint8 * 3),
                ("cfl_coeff", C.c_float), ("lambda_", C.c_float), ("mu", C.c_float)]


class NeoHookeanElasticityC(C.Structure):
    _fields_ = [("cfl_coeff", C.c_float), ("lambda_", C.c_float), ("mu", C.c_float)]


class MonaghanSphEosC(C.Structure):
    _fields_ = [("pressure0", C.c_float), ("gamma", C.c_uint32), ("viscosity", C.c_float),
                ("max_neg_pressure", C.c_float)]


class _ConstitutiveU(C.Union):
    _fields_ = [("corotated", CorotatedLinearElasticityC), ("neo_hookean", NeoHookeanElasticityC),
                ("monaghan", MonaghanSphEosC), ("custom_id", C.c_uint32)]


class ConstitutiveModelC(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("u", _ConstitutiveU)]


class DruckerPragerPlasticityC(C.Structure):
    _fields_ = [("h0", C.c_float), ("h1", C.c_float), ("h2", C.c_float), ("h3", C.c_float),
                ("lambda_", C.c_float), ("mu", C.c_float),
                ("only_active_when_failed", C.c_uint8), ("_pad", C.c_uint8 * 3),
                ("volume_correction", C.c_float)]


class NaccPlasticityC(C.Structure):
    _fields_ = [("mu", C.c_float), ("kappa", C.c_float), ("hardening_enabled", C.c_uint8),
                ("_pad", C.c_uint8 * 3), ("hardening_factor", C.c_float), ("cohesion", C.c_float),
                ("friction", C.c_float)]


class RankinePlasticityC(C.Structure):
    _fields_ = [("mu", C.c_float), ("lambda_", C.c_float), ("tensile_strength", C.c_float),
                ("softening_rate", C.c_float)]


class SnowPlasticityC(C.Structure):
    _fields_ = [("min_epsilon", C.c_float), ("max_epsilon", C.c_float), ("hardening_coeff", C.c_float)]


class _PlasticU(C.Union):
    _fields_ = [("drucker_prager", DruckerPragerPlasticityC), ("nacc", NaccPlasticityC),
                ("rankine", RankinePlasticityC), ("snow", SnowPlasticityC), ("custom_id", C.c_uint32)]


class PlasticModelC(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("u", _PlasticU)]


class MaximumStressFailureC(C.Structure):
    _fields_ = [("max_principal_stress", C.c_float), ("max_shear_stress", C.c_float)]


class _FailureU(C.Union):
    _fields_ = [("maximum_stress", MaximumStressFailureC), ("custom_id", C.c_uint32)]


class FailureModelC(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("u", _FailureU)]


class ParticleModelC(C.Structure):
    _fields_ = [("constitutive_model", ConstitutiveModelC), ("has_plastic_model", C.c_uint32),
                ("plastic_model", PlasticModelC), ("has_failure_model", C.c_uint32),
                ("failure_model", FailureModelC)]


CONSTITUTIVE_COROTATED_LINEAR, CONSTITUTIVE_NEO_HOOKEAN, CONSTITUTIVE_EOS_MONAGHAN_SPH, CONSTITUTIVE_CUSTOM = 0, 1, 2, 3
PLASTIC_DRUCKER_PRAGER, PLASTIC_NACC, PLASTIC_RANKINE, PLASTIC_SNOW, PLASTIC_CUSTOM = 0, 1, 2, 3, 4
FAILURE_MAXIMUM_STRESS, FAILURE_CUSTOM = 0, 1


class CuboidColliderC(C.Structure):
    _fields_ = [("half_extents", C.c_float * 3), ("rotation", C.c_float * 4),
                ("translation", C.c_float * 3), ("friction", C.c_float)]


class PolylineColliderC(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("vertices", C.POINTER(C.c_float)), ("rotation", C.c_float * 4), ("translation", C.c_float * 3), ("friction", C.c_float)]


def polyline(vertices_xy, translation=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0), friction=0.5):
    """`ColliderBuilder::polyline(vertices, None)` as the reference's GPU path reads it (gpu_collider.rs:99-201): a closed counter-clockwise polygon
    in the local x-y plane, extruded along z.  `vertices_xy`: (n, 2) or (n, 3)."""
    v = np.asarray(vertices_xy, dtype=np.float32)
    if v.shape[1] == 2:
        v = np.concatenate([v, np.zeros((len(v), 1), np.float32)], axis=1)
    v = np.ascontiguousarray(v, dtype=np.float32)
    c = PolylineColliderC()
    c.num_vertices = len(v)
    c.vertices = v.ctypes.data_as(C.POINTER(C.c_float))
    c._keep = v
    c.rotation[:] = [float(x) for x in rotation]
    c.translation[:] = [float(x) for x in translation]
    c.friction = float(friction)
    return c


class ColliderOptionsC(C.Structure):
    """`CudaColliderOptions` (src/cuda/cuda_collider_set.rs:22-38) without the handle: one per collider of a kind, in set_* order."""
    _fields_ = [("grid_boundary_handling", C.c_int32), ("penalty_stiffness", C.c_float), ("flip_interior", C.c_uint8), ("_pad", C.c_uint8 * 3)]


COLLIDER_HANDLING_FROM_PARAMS = -1


def collider_options(grid_boundary_handling=COLLIDER_HANDLING_FROM_PARAMS, penalty_stiffness=0.0, flip_interior=False):
    o = ColliderOptionsC()
    o.grid_boundary_handling, o.penalty_stiffness, o.flip_interior = int(grid_boundary_handling), float(penalty_stiffness), int(bool(flip_interior))
    return o


class HeightFieldColliderC(C.Structure):
    _fields_ = [("nrows", C.c_uint32), ("ncols", C.c_uint32), ("heights", C.POINTER(C.c_float)), ("scale", C.c_float * 3),
                ("rotation", C.c_float * 4), ("translation", C.c_float * 3), ("friction", C.c_float),
                ("below_is_solid", C.c_uint8), ("_pad", C.c_uint8 * 3)]


def heightfield(heights, scale, translation=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0), friction=0.5, below_is_solid=False):
    """`ColliderBuilder::heightfield(heights, scale).translation(..)`: heights[i, j], rows along z, columns along x."""
    hts = np.asfortranarray(np.asarray(heights, dtype=np.float32))      # nalgebra DMatrix is column-major
    c = HeightFieldColliderC()
    c.nrows, c.ncols = hts.shape
    c._keep = hts                                                       # keep the buffer alive
    c.heights = hts.ctypes.data_as(C.POINTER(C.c_float))
    c.scale[:] = [float(x) for x in scale]
    c.rotation[:] = [float(x) for x in rotation]
    c.translation[:] = [float(x) for x in translation]
    c.friction = friction
    c.below_is_solid = int(below_is_solid)
    return c


class TriMeshColliderC(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("vertices", C.POINTER(C.c_float)), ("num_triangles", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("rotation", C.c_float * 4), ("translation", C.c_float * 3), ("friction", C.c_float),
                ("oriented", C.c_uint8), ("flip_interior", C.c_uint8), ("_pad", C.c_uint8 * 2)]


def trimesh(vertices, indices, translation=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0), friction=0.5, oriented=False, flip_interior=False):
    """`ColliderBuilder::trimesh(vertices, indices)` (+ TriMeshFlags::ORIENTED through `oriented`)."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
    c = TriMeshColliderC()
    c._keep = (v, i)
    c.num_vertices, c.num_triangles = len(v), len(i)
    c.vertices = v.ctypes.data_as(C.POINTER(C.c_float))
    c.indices = i.ctypes.data_as(C.POINTER(C.c_uint32))
    c.rotation[:] = [float(x) for x in rotation]
    c.translation[:] = [float(x) for x in translation]
    c.friction = friction
    c.oriented, c.flip_interior = int(oriented), int(flip_interior)
    return c


class TimingsC(C.Structure):
    _fields_ = [("dt", C.c_float), ("num_substeps", C.c_uint32), ("num_resorts", C.c_uint32),
                ("grid_resize_and_sort_ms", C.c_float), ("reset_grid_ms", C.c_float),
                ("estimate_timestep_ms", C.c_float), ("p2g_ms", C.c_float), ("grid_update_ms", C.c_float),
                ("g2p_ms", C.c_float), ("halo_ms", C.c_float), ("total_ms", C.c_float)]


class SubstepTimingsC(C.Structure):
    """One entry of `CudaTimestepTimings::substeps` (cuda_mpm_pipeline.rs:38-85): SparklSubstepTimings."""
    _fields_ = [("dt", C.c_float), ("grid_resize_and_sort_ms", C.c_float), ("estimate_timestep_ms", C.c_float), ("p2g_ms", C.c_float),
                ("grid_update_ms", C.c_float), ("g2p_ms", C.c_float), ("halo_ms", C.c_float), ("total_ms", C.c_float)]


class KernelClockC(C.Structure):
    _fields_ = [("p2g_ms", C.c_double), ("grid_update_ms", C.c_double), ("g2p_ms", C.c_double),
                ("sort_ms", C.c_double), ("dt_ms", C.c_double), ("halo_ms", C.c_double),
                ("substeps", C.c_uint64), ("resorts", C.c_uint64), ("sampled_substeps", C.c_uint64), ("g2p2g_ms", C.c_double),
                ("p2g_launches", C.c_uint64), ("grid_update_launches", C.c_uint64), ("g2p_launches", C.c_uint64), ("g2p2g_launches", C.c_uint64),
                ("reorder_ms", C.c_double), ("reorder_launches", C.c_uint64),
                ("p2g_mass_ms", C.c_double), ("grid_update_mass_ms", C.c_double), ("g2p_volume_ms", C.c_double),
                ("p2g_mass_launches", C.c_uint64), ("grid_update_mass_launches", C.c_uint64), ("g2p_volume_launches", C.c_uint64),
                ("noop_ms", C.c_double), ("noop_launches", C.c_uint64)]


class GridCheckC(C.Structure):
    _fields_ = [("bad_particle_ids", C.c_uint64), ("blocks_not_in_hash", C.c_uint64), ("particles_outside_their_block", C.c_uint64),
                ("missing_neighbour_blocks", C.c_uint64), ("coverage_errors", C.c_uint64), ("node_blocks", C.c_uint64), ("particle_blocks", C.c_uint64)]


class ReorderStatsC(C.Structure):
    _fields_ = [("last_heads", C.c_uint32), ("base_heads", C.c_uint32), ("reorders", C.c_uint32), ("policy_rebins", C.c_uint32)]


class PipelineDescC(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("cell_width", C.c_float), ("domain_log2_bytes", C.c_uint32),
                ("stream", C.c_void_p)]


def ptr(arr, dtype):
    """numpy array -> void* after checking dtype/contiguity (None -> NULL)."""
    if arr is None:
        return None
    assert arr.dtype == dtype and arr.flags["C_CONTIGUOUS"], (arr.dtype, dtype)
    return arr.ctypes.data_as(C.c_void_p)
