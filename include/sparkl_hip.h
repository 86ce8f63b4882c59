/*
 * sparkl_hip.h -- C ABI of libsparkl_hip.so, the MI355X (gfx950) MPM hot path.
 *
 * This is the drop-in boundary for sparkl3d's (dim3, f32) MPM substep:
 *   sort/bin -> [fluid volume recompute] -> adaptive dt -> P2G (+stress) ->
 *   grid update -> G2P (+F update, plasticity)
 * i.e. what `MlsSolver::step` does on the CPU
 *   (reference: src/dynamics/solver/mls_solver.rs:15-139)
 * exposed through the *shape* of the reference's GPU host API
 *   (reference: src/cuda/cuda_mpm_pipeline.rs:246-271, src/cuda/cuda_particle_set.rs:12-177,
 *    src/cuda/cuda_particle_model_set.rs:37-84, src/cuda/cuda_collider_set.rs:46-127).
 *
 * Every struct below is layout-identical to the `#[repr(C)]` Rust type it cites,
 * so a Rust host can pass `&[ParticlePosition]`, `&SolverParameters`, ... directly.
 * All functions return 0 on success or a SPARKL_ERR_* / hipError-derived code;
 * `sparkl_hip_last_error_string()` gives the text.  Nothing here aborts.
 * A handle is NOT thread-safe (one caller thread at a time, as in the reference,
 * src/cuda/cuda_mpm_pipeline.rs:235-237).
 */
#ifndef SPARKL_HIP_H
#define SPARKL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors */
enum {
    SPARKL_OK = 0,
    SPARKL_ERR_INVALID_ARGUMENT = 1,
    SPARKL_ERR_UNSUPPORTED = 2,      /* e.g. Core*Model::Custom(u32): device code cannot cross a C ABI */
    SPARKL_ERR_NO_DEVICE = 3,        /* no HIP device / HIP runtime failure at create */
    SPARKL_ERR_HIP = 4,              /* a HIP call failed; see last_error_string */
    SPARKL_ERR_NAN_TIMESTEP = 5,     /* reference panics here (timestep_estimator.rs:46) */
    SPARKL_ERR_OUT_OF_MEMORY = 6,
    SPARKL_ERR_INTERNAL = 7
};

/* ------------------------------------------- particle components (SoA, f32)
 * reference: src_core/dynamics/particle.rs:18-219 */
typedef struct { float point[3]; } SparklParticlePosition;           /* :18-21, 12 B */
typedef struct { float vector[3]; } SparklParticleVelocity;          /* :54-57, 12 B */
typedef struct {                                                     /* :64-72, 52 B */
    float mass;
    float volume0;
    float radius0;
    float deformation_gradient[9];            /* column-major (nalgebra Matrix3).  The library keeps F - I per particle (exact on upload for
                                               * 0.5 <= F_kk <= 2, one rounding when F is formed for a read-back): DESIGN.md 5 */
    float plastic_deformation_gradient_det;
} SparklParticleVolume;
typedef struct {                                                     /* :114-122, 24 B */
    uint8_t failed;
    uint8_t is_static;
    uint8_t kinematic_vel_enabled;
    uint8_t _pad0;
    float kinematic_vel[3];
    uint64_t model_index;                     /* Rust usize */
} SparklParticleStatus;
typedef struct { float phase; float psi_pos; } SparklParticlePhase;  /* :138-143, 8 B */
typedef struct {                                                     /* :201-208, 16 B */
    float nacc_alpha;
    float plastic_hardening;
    float elastic_hardening;
    float log_vol_gain;
} SparklParticleData;
/* Optional component (particle.rs:180-195; `Particle::{crack_propagation_factor, crack_threshold}` on the CPU path,
 * src/dynamics/particle.rs:29-30): only read under SPARKL_DAMAGE_MODIFIED_EIGENEROSION.  Default (0, f32::MAX). */
typedef struct { float crack_propagation_factor; float crack_threshold; } SparklParticleFracture;   /* 8 B */
/* CPU-path-only state: `Particle::velocity_gradient` (src/dynamics/particle.rs:16).
 * The reference CUDA path never stores it (fused g2p2g); the CPU ordering needs it
 * between G2P and the next P2G.  Column-major. */
typedef struct { float m[9]; } SparklVelocityGradient;               /* 36 B */

/* --------------------------------------------------------- solver parameters
 * reference: src_core/dynamics/solver/solver_parameters.rs:7-68 */
enum { SPARKL_DOFS_LOCK_NONE = 0, SPARKL_DOFS_LOCK_X = 1, SPARKL_DOFS_LOCK_Y = 2, SPARKL_DOFS_LOCK_Z = 4 };
enum { SPARKL_BOUNDARY_STICK = 0, SPARKL_BOUNDARY_FRICTION = 1, SPARKL_BOUNDARY_FRICTION_Z_UP = 2, SPARKL_BOUNDARY_NONE = 3 };
enum { SPARKL_DAMAGE_CDMPM = 0, SPARKL_DAMAGE_EIGENEROSION = 1, SPARKL_DAMAGE_MODIFIED_EIGENEROSION = 2, SPARKL_DAMAGE_NONE = 3 };

typedef struct {                                                     /* :42-54, 28 B */
    float dt;
    float max_substep_dt;
    uint32_t max_num_substeps;
    int32_t boundary_handling;
    int32_t damage_model;
    uint8_t force_fluids_volume_recomputation;
    uint8_t enable_boundary_particle_projection;
    uint8_t stop_after_one_substep;
    uint8_t _pad0;
    uint32_t simulation_dofs;
} SparklSolverParams;

/* -------------------------------------------------------------------- models
 * `#[repr(C)] enum` with payload == { tag; union } in C.
 * reference: src_core/dynamics/models/constitutive_model.rs:14-23,
 *            plastic_model.rs:5-15, failure_model.rs:4-11 */
typedef struct { uint8_t split_stress_on_failure; uint8_t _pad[3]; float cfl_coeff; float lambda; float mu; }
    SparklCorotatedLinearElasticity;        /* elasticity_corotated_linear.rs:8-16 */
typedef struct { float cfl_coeff; float lambda; float mu; }
    SparklNeoHookeanElasticity;             /* elasticity_neo_hookean.rs:7-14 */
typedef struct { float pressure0; uint32_t gamma; float viscosity; float max_neg_pressure; }
    SparklMonaghanSphEos;                   /* eos_monaghan_sph.rs:7-16 */
enum {
    SPARKL_CONSTITUTIVE_COROTATED_LINEAR = 0,
    SPARKL_CONSTITUTIVE_NEO_HOOKEAN = 1,
    SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH = 2,
    SPARKL_CONSTITUTIVE_CUSTOM = 3
};
typedef struct {
    uint32_t tag;
    union {
        SparklCorotatedLinearElasticity corotated;
        SparklNeoHookeanElasticity neo_hookean;
        SparklMonaghanSphEos monaghan;
        uint32_t custom_id;
    } u;
} SparklConstitutiveModel;                  /* 20 B */

typedef struct {                            /* plasticity_drucker_prager.rs:6-19, 32 B */
    float h0, h1, h2, h3;
    float lambda, mu;
    uint8_t only_active_when_failed; uint8_t _pad[3];
    float volume_correction;
} SparklDruckerPragerPlasticity;
typedef struct { float mu, kappa; uint8_t hardening_enabled; uint8_t _pad[3]; float hardening_factor, cohesion, friction; }
    SparklNaccPlasticity;                   /* plasticity_nacc.rs:8-19 */
typedef struct { float mu, lambda, tensile_strength, softening_rate; }
    SparklRankinePlasticity;                /* plasticity_rankine.rs:7-16 */
typedef struct { float min_epsilon, max_epsilon, hardening_coeff; }
    SparklSnowPlasticity;                   /* plasticity_snow.rs:7-15 */
enum {
    SPARKL_PLASTIC_DRUCKER_PRAGER = 0,
    SPARKL_PLASTIC_NACC = 1,
    SPARKL_PLASTIC_RANKINE = 2,
    SPARKL_PLASTIC_SNOW = 3,
    SPARKL_PLASTIC_CUSTOM = 4
};
typedef struct {
    uint32_t tag;
    union {
        SparklDruckerPragerPlasticity drucker_prager;
        SparklNaccPlasticity nacc;
        SparklRankinePlasticity rankine;
        SparklSnowPlasticity snow;
        uint32_t custom_id;
    } u;
} SparklPlasticModel;                       /* 36 B */

typedef struct { float max_principal_stress, max_shear_stress; } SparklMaximumStressFailure;
enum { SPARKL_FAILURE_MAXIMUM_STRESS = 0, SPARKL_FAILURE_CUSTOM = 1 };
typedef struct {
    uint32_t tag;
    union { SparklMaximumStressFailure maximum_stress; uint32_t custom_id; } u;
} SparklFailureModel;                       /* 12 B */

/* `ParticleModel{constitutive_model, plastic_model: Option, failure_model: Option}`
 * (src/dynamics/particle_model.rs:14-18) after `to_core_model()`.  Rust `Option` has no
 * C layout, hence the explicit has_* bytes. */
typedef struct {
    SparklConstitutiveModel constitutive_model;
    uint32_t has_plastic_model;
    SparklPlasticModel plastic_model;
    uint32_t has_failure_model;
    SparklFailureModel failure_model;
} SparklParticleModel;

/* ----------------------------------------------------------------- colliders
 * First slice of `CudaColliderSet::from_collider_set` (src/cuda/cuda_collider_set.rs:46-127)
 * and of the CPU `RigidWorld{colliders}` loop (src/dynamics/solver/grid_update.rs:77-128):
 * cuboids only.  `position` is a Rapier isometry: unit quaternion (i,j,k,w) + translation. */
typedef struct {
    float half_extents[3];
    float rotation[4];      /* (i, j, k, w) */
    float translation[3];
    float friction;         /* collider.friction(), grid_update.rs:113 */
} SparklCuboidCollider;

/* Heightfield (rapier `ColliderBuilder::heightfield(heights, scale)`, examples3d/sand3.rs:23-37): parry3d 0.13 layout --
 * heights is an nrows x ncols nalgebra DMatrix (column-major: (i, j) at i + nrows * j), rows run along local z, columns
 * along local x, both spanning [-0.5, 0.5] * scale; every cell is split into the triangles (p00, p10, p01) and
 * (p10, p11, p01).  The CPU path projects on the triangle soup and never reports `is_inside`
 * (grid_update.rs:78-86); `below_is_solid` != 0 selects the CUDA path's rule instead
 * (src_kernels/gpu_collider.rs:69-91: inside = within the x-z extent and not above the surface). */
typedef struct {
    uint32_t nrows, ncols;
    const float* heights;
    float scale[3];
    float rotation[4];      /* (i, j, k, w) */
    float translation[3];
    float friction;
    uint8_t below_is_solid;
    uint8_t _pad[3];
} SparklHeightFieldCollider;

/* rapier `ColliderBuilder::trimesh(vertices, indices)` (parry3d TriMesh), the third shape of
 * CudaColliderSet::from_collider_set (cuda_collider_set.rs:88-102).  Vertices in the collider's local frame.
 * `oriented` = TriMeshFlags::ORIENTED: parry then decides inside / outside from the pseudo-normal of the closest feature
 * (angle-weighted vertex normals, summed face normals on edges); without it the mesh is a two-sided surface and
 * is_inside is never set, which is what the default rapier trimesh gives the CPU path (grid_update.rs:78-82).
 * `flip_interior` as CudaColliderOptions::flip_interior (only meaningful with `oriented`). */
typedef struct {
    uint32_t num_vertices;
    const float* vertices;        /* 3 * num_vertices */
    uint32_t num_triangles;
    const uint32_t* indices;      /* 3 * num_triangles */
    float rotation[4];            /* (i, j, k, w) */
    float translation[3];
    float friction;
    uint8_t oriented;
    uint8_t flip_interior;
    uint8_t _pad[2];
} SparklTriMeshCollider;

/* rapier `ColliderBuilder::polyline(vertices, None)` as CudaColliderSet::from_collider_set takes it (cuda_collider_set.rs:102-117,
 * src_kernels/gpu_collider.rs:99-201): a CLOSED polygon, counter-clockwise in the collider's local x-y plane (vertex i is joined to vertex
 * (i + 1) % n), extruded along local z; the inside of the polygon is solid.  The CPU path (parry's open Polyline) never reports is_inside;
 * this is the GPU path's shape, the only one the reference defines for MPM boundaries. */
typedef struct {
    uint32_t num_vertices;
    const float* vertices;        /* 3 * num_vertices, local frame */
    float rotation[4];            /* (i, j, k, w) */
    float translation[3];
    float friction;
} SparklPolylineCollider;

/* `CudaColliderOptions` (src/cuda/cuda_collider_set.rs:22-38): what the reference's GPU path lets a caller choose PER COLLIDER, beside the
 * collider itself.  Given to sparkl_hip_colliders_set_*_options, one record per collider of that kind, in the order of the matching
 * sparkl_hip_colliders_set_* call (which resets every collider of its kind to the defaults).
 *   grid_boundary_handling  SPARKL_BOUNDARY_* for this collider, or SPARKL_COLLIDER_HANDLING_FROM_PARAMS (-1, the default): whatever
 *                           SolverParameters::boundary_handling says -- the CPU path's single switch (grid_update.rs:84-128), which is what
 *                           every collider did until round 5.  SPARKL_BOUNDARY_NONE takes the collider out of the grid update.
 *                           As soon as ONE collider of the set carries a handling of its own, the whole set follows the reference's GPU grid
 *                           update (src_kernels/cuda/grid_update.rs:97-190), the only place these options exist: of all colliders whose rule is
 *                           not NONE only the one whose projection of the node is CLOSEST acts, and the collision normal is the normalised
 *                           central-difference gradient of the signed distance to all of them (grid_update.rs:54-95), not the direction of the
 *                           projection; SPARKL_BOUNDARY_FRICTION_Z_UP applies friction only where that normal has z >= 0.  Without any such
 *                           option the CPU path's rule stands: every collider acts in turn (grid_update.rs:84-128).
 *   penalty_stiffness       force (proj - x) * stiffness on a particle found inside the collider (src_kernels/cuda/particle_updater.rs:200-225).
 *                           The reference compiles this force OUT (`if false`, particle_updater.rs:204): the value is stored and changes nothing,
 *                           and that is the default here too.  sparkl_hip_colliders_enable_penalty_force(p, 1) (or SPARKL_HIP_PENALTY_FORCE=1 in
 *                           the environment at create) applies the intended rule: the force enters the momentum the particle scatters next,
 *                           m v + f dt (src_kernels/cuda/g2p2g.rs:286-287).
 *   flip_interior           heightfields (with below_is_solid) and oriented triangle meshes: swaps inside and outside
 *                           (src_kernels/gpu_collider.rs:58-92).  For a triangle mesh it overrides SparklTriMeshCollider::flip_interior. */
enum { SPARKL_COLLIDER_HANDLING_FROM_PARAMS = -1 };
typedef struct {
    int32_t grid_boundary_handling;
    float penalty_stiffness;
    uint8_t flip_interior;
    uint8_t _pad[3];
} SparklColliderOptions;

/* ------------------------------------------------------------------- timings
 * reference: CudaTimestepTimings / CudaSubstepTimings / CudaDeviceTimings
 * (src/cuda/cuda_mpm_pipeline.rs:28-85).  Milliseconds from hipEvents, summed over the
 * substeps of the last `sparkl_hip_pipeline_step` call.  g2p2g is split in p2g + g2p. */
typedef struct {
    float dt;                       /* simulated time of the call */
    uint32_t num_substeps;
    uint32_t num_resorts;           /* how many substeps re-binned the particles */
    float grid_resize_and_sort_ms;
    float reset_grid_ms;            /* always 0: tiles are written whole, nothing to reset */
    float estimate_timestep_ms;
    float p2g_ms;
    float grid_update_ms;
    float g2p_ms;
    float halo_ms;                  /* pack + merge of ghost planes (multi-GPU slabs) */
    float total_ms;
} SparklTimings;

/* One entry of `CudaTimestepTimings::substeps` (`CudaSubstepTimings` + the one `CudaDeviceTimings` of a single-device step,
 * src/cuda/cuda_mpm_pipeline.rs:38-85): step length and device milliseconds of ONE substep of the last sparkl_hip_pipeline_step call that was
 * given a SparklTimings.  A re-bin and the estimator pass are booked on the substep they precede; total_ms is the sum of the others
 * (device time of the substep's own launches: gaps between launches are in SparklTimings::total_ms only). */
typedef struct {
    float dt;
    float grid_resize_and_sort_ms;
    float estimate_timestep_ms;
    float p2g_ms;
    float grid_update_ms;
    float g2p_ms;
    float halo_ms;
    float total_ms;
} SparklSubstepTimings;

/* ------------------------------------------------------------------ pipeline */
typedef struct SparklPipeline SparklPipeline;

typedef struct {
    int32_t device_id;      /* HIP device ordinal */
    float cell_width;       /* SpGrid::new(cell_width), src/geometry/sp_grid.rs:168 */
    /* log2 of the virtual SPGrid size in bytes; 30 reproduces the intended 1 GiB map
     * (src/geometry/sp_grid.rs:175) => valid cell coords 0..255 per axis.
     * Particles whose neighbourhood leaves the domain are flagged `failed`
     * (src/dynamics/particle_set.rs:129-136).  Allowed: 30, 33, 36, 39. */
    uint32_t domain_log2_bytes;
    void* stream;           /* hipStream_t to run on, or NULL for a library-owned stream */
} SparklPipelineDesc;

/* CudaMpmPipeline::new + load_module + CudaSparseGrid::new
 * (cuda_mpm_pipeline.rs:247-260, cuda_sparse_grid.rs:49-90) */
int sparkl_hip_pipeline_create(const SparklPipelineDesc* desc, SparklPipeline** out);
void sparkl_hip_pipeline_destroy(SparklPipeline* p);

/* CudaParticleSet::from_particles (cuda_particle_set.rs:33-71) + the shared ParticleData
 * attribute buffer (cuda_particle_model.rs:14-29).  `velocity_gradients` may be NULL (zeros,
 * as `Particle::new`, src/dynamics/particle.rs:79).  Host arrays are only borrowed. */
int sparkl_hip_particles_upload(SparklPipeline* p, uint64_t n,
                                const SparklParticleStatus* status,
                                const SparklParticlePosition* pos,
                                const SparklParticleVelocity* vel,
                                const SparklParticleVolume* volume,
                                const SparklParticlePhase* phase,
                                const SparklParticleData* data,
                                const SparklVelocityGradient* velocity_gradients);
/* CudaParticleSet::append (cuda_particle_set.rs:135-177, cuda_vec.rs:56-81): incremental -- the buffers grow in place and
 * only the new records cross PCIe; caller indices of the new particles continue after the old ones. */
int sparkl_hip_particles_append(SparklPipeline* p, uint64_t n,
                                const SparklParticleStatus* status,
                                const SparklParticlePosition* pos,
                                const SparklParticleVelocity* vel,
                                const SparklParticleVolume* volume,
                                const SparklParticlePhase* phase,
                                const SparklParticleData* data,
                                const SparklVelocityGradient* velocity_gradients);
/* CudaParticleSet::remove_range / truncate / len (cuda_particle_set.rs:73-133, cuda_vec.rs:83-110): device side; later
 * caller indices move down like a Vec's.  Triggers a re-bin. */
int sparkl_hip_particles_remove_range(SparklPipeline* p, uint64_t start, uint64_t end);
int sparkl_hip_particles_truncate(SparklPipeline* p, uint64_t new_len);
uint64_t sparkl_hip_particles_len(const SparklPipeline* p);
/* ParticleFracture of the resident particles, caller order, n == particles_len.  Travels with the particles through re-bins;
 * appended particles get the default.  Read by DamageModel::ModifiedEigenerosion only (grid_to_particle.rs:65-77):
 * the psi_mass / psi_momentum node transfer of particle_to_grid.rs:62-86 runs as three extra kernels per substep then. */
int sparkl_hip_particles_set_fracture(SparklPipeline* p, const SparklParticleFracture* fracture, uint64_t n);
int sparkl_hip_particles_read_fracture(SparklPipeline* p, SparklParticleFracture* out, uint64_t n);
int sparkl_hip_particles_has_fracture(const SparklPipeline* p);   /* 1 once a component was uploaded (it also selects the unfused schedule, see below) */

/* CudaParticleSet::read_* (cuda_particle_set.rs:91-121).  Arrays come back in CALLER order
 * (the library keeps particles physically block-sorted on the device).  `n` must equal len. */
int sparkl_hip_particles_read_positions(SparklPipeline* p, SparklParticlePosition* out, uint64_t n);
int sparkl_hip_particles_read_velocities(SparklPipeline* p, SparklParticleVelocity* out, uint64_t n);
int sparkl_hip_particles_read_volumes(SparklPipeline* p, SparklParticleVolume* out, uint64_t n);
int sparkl_hip_particles_read_statuses(SparklPipeline* p, SparklParticleStatus* out, uint64_t n);
int sparkl_hip_particles_read_phases(SparklPipeline* p, SparklParticlePhase* out, uint64_t n);
int sparkl_hip_particles_read_data(SparklPipeline* p, SparklParticleData* out, uint64_t n);
int sparkl_hip_particles_read_velocity_gradients(SparklPipeline* p, SparklVelocityGradient* out, uint64_t n);
/* read_sorted_particle_ids: caller index of the particle stored in each sorted slot */
int sparkl_hip_particles_read_sorted_ids(SparklPipeline* p, uint32_t* out, uint64_t n);

/* Per-frame readback for rendering (testbed_plugin.rs:533-588: positions, velocities and phases are copied into
 * `LockedBuffer`s every frame).  `sparkl_hip_host_alloc` = LockedBuffer::uninitialized (page-locked host memory).
 * `readback_begin` snapshots the state after everything enqueued so far, in caller order, and returns at once:
 * the device-to-host copies run on their own stream, so the next `pipeline_step` overlaps them
 * (10 M particles x 32 B = 320 MB per frame).  Any of pos / vel / phase may be NULL.  The host arrays may be read
 * after `readback_wait`; a second `readback_begin` queues behind the first.  Pageable host pointers work too
 * (the copy then degrades to a staged one). */
int sparkl_hip_host_alloc(void** out, uint64_t bytes);
int sparkl_hip_host_free(void* ptr);
int sparkl_hip_readback_begin(SparklPipeline* p, SparklParticlePosition* pos, SparklParticleVelocity* vel, SparklParticlePhase* phase, uint64_t n);
int sparkl_hip_readback_wait(SparklPipeline* p);

/* CudaParticleModelSet::from_model_set / ensure_models_are_uploaded
 * (cuda_particle_model_set.rs:37-84): call again whenever a model changed. */
int sparkl_hip_models_set(SparklPipeline* p, uint32_t n, const SparklParticleModel* models);
/* CudaColliderSet::from_collider_set (cuboids) */
int sparkl_hip_colliders_set_cuboids(SparklPipeline* p, uint32_t n, const SparklCuboidCollider* colliders);
/* CudaColliderSet::from_collider_set (heightfields); applied after the cuboids.  Heights are copied. */
int sparkl_hip_colliders_set_heightfields(SparklPipeline* p, uint32_t n, const SparklHeightFieldCollider* colliders);
/* CudaColliderSet::from_collider_set (triangle meshes); applied after the heightfields.  The mesh is copied and indexed by
 * a uniform grid of its bounding box built on the host. */
int sparkl_hip_colliders_set_trimeshes(SparklPipeline* p, uint32_t n, const SparklTriMeshCollider* colliders);
/* CudaColliderSet::from_collider_set's `options: Vec<CudaColliderOptions>` (cuda_collider_set.rs:46-58), per collider kind: n must equal the
 * number of colliders of that kind currently set (options == NULL with n == 0 is fine).  Defaults: see SparklColliderOptions. */
int sparkl_hip_colliders_set_cuboid_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* options);
int sparkl_hip_colliders_set_heightfield_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* options);
int sparkl_hip_colliders_set_trimesh_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* options);
/* CudaColliderSet::from_collider_set (polylines); applied after the triangle meshes.  Vertices are copied. */
int sparkl_hip_colliders_set_polylines(SparklPipeline* p, uint32_t n, const SparklPolylineCollider* colliders);
int sparkl_hip_colliders_set_polyline_options(SparklPipeline* p, uint32_t n, const SparklColliderOptions* options);
/* SparklColliderOptions::penalty_stiffness acts only behind this switch: the reference has the force under `if false`
 * (src_kernels/cuda/particle_updater.rs:203-221), so by default (enable = 0) a stiffness is stored and ignored, as there. */
int sparkl_hip_colliders_enable_penalty_force(SparklPipeline* p, int enable);

/* MpmPipeline::step / CudaMpmPipeline::step
 * (src/pipelines/mpm_pipeline.rs:15-42, src/cuda/cuda_mpm_pipeline.rs:262-271):
 * advances by params->dt in adaptive substeps (CPU-path rules, mls_solver.rs:33-59,125);
 * returns the simulated time (params->dt, or the substep dt if stop_after_one_substep). */
int sparkl_hip_pipeline_step(SparklPipeline* p, const SparklSolverParams* params,
                             const float gravity[3], SparklTimings* timings_or_null,
                             float* simulated_time);

/* `timings.substeps` of the last sparkl_hip_pipeline_step call that was given a SparklTimings: *n_out = number of substeps of that call,
 * the first min(cap, *n_out) records are written to `out` (cap = 0 just asks for the count). */
int sparkl_hip_pipeline_step_substep_timings(SparklPipeline* p, SparklSubstepTimings* out, uint32_t cap, uint32_t* n_out);

/* ---- extensions (no reference counterpart; used by bench / parity tests) ---- */
/* Run `nsubsteps` substeps of fixed length `dt` with NO host synchronisation inside
 * (the dt estimator is skipped; equivalent to the reference when max_substep_dt == dt
 * is the binding bound).  Only params->{boundary_handling, simulation_dofs,
 * force_fluids_volume_recomputation, enable_boundary_particle_projection, damage_model} are read.
 * damage_model: SPARKL_DAMAGE_NONE and SPARKL_DAMAGE_MODIFIED_EIGENEROSION are implemented here and in
 * sparkl_hip_pipeline_step (the latter runs its psi_mass / psi_momentum sweep only while a ParticleFracture
 * component is uploaded -- sparkl_hip_particles_set_fracture -- and turns the fused G2P->P2G schedule off);
 * SPARKL_DAMAGE_CDMPM / SPARKL_DAMAGE_EIGENEROSION return SPARKL_ERR_UNSUPPORTED everywhere, and the
 * sparkl_hip_group_substeps accepts both as well (the psi planes and, with force_fluids_volume_recomputation, the mass planes of
 * the cuts travel in extra halo rounds); the split-phase entry points (sparkl_hip_substep_begin ...) accept SPARKL_DAMAGE_NONE only.  Substeps are enqueued in batches; the host stays a few substeps
 * ahead of the GPU (it polls two words of pinned memory the kernels write: the step counter and the re-bin request) and reads the control block
 * (80 bytes) back once per batch (a stream sync), so the call returns when the last batch has been read. */
int sparkl_hip_pipeline_substeps(SparklPipeline* p, const SparklSolverParams* params,
                                 const float gravity[3], float dt, uint32_t nsubsteps);
int sparkl_hip_pipeline_sync(SparklPipeline* p);

/* Re-order policy.  Particles keep their memory slots between re-bins, so lanes that shared a cell drift apart and the
 * scatter's lane pre-reduction degrades (k_g2p2g: 0.57 -> 1.0 ms over ~300 substeps of the collapsing 10M column).
 * Every scatter counts its heads (fused run-task scatter: the distinct cells of each wave -- it groups a wave's lanes by cell
 * itself, so only the spread of a wave over cells costs; DPP scatters of the fluid sweep and the standalone P2G: lane-group
 * heads); the heads in excess of the count right after the last re-bin are summed over the substeps since, and
 *   - past `reorder_limit` the particle blocks whose own count has grown by an eighth are re-sorted by current cell in
 *     place (device only; every block under the DPP scatters, ~0.6 ms at 10M),
 *   - past `rebin_limit` a full re-bin is requested through the same abort chain a margin violation uses.
 * Limits are in heads (the cost of the remedy / the cost of one excess head).  0 = off, < 0 = built-in default
 * (reorder: 0.2 heads per particle, 0.6 when a fluid model is present; full re-bin: off).  Physical results depend on the policy only at
 * rounding level: the order of the f32 sums, and WHICH particles share a 64-lane wave -- the return map picks its series tier / eigenbasis
 * path per wave (wave-uniform branches), and the paths agree to ~1e-7 of the strain, not bit for bit. */
int sparkl_hip_reorder_policy_set(SparklPipeline* p, float reorder_limit, float rebin_limit);
typedef struct {
    uint32_t last_heads, base_heads;      /* run heads of the last scatter / right after the last re-bin */
    uint32_t reorders, policy_rebins;     /* in-place reorders and full re-bins the policy has triggered */
} SparklReorderStats;
int sparkl_hip_reorder_policy_stats(SparklPipeline* p, SparklReorderStats* out);

/* Structure check of the binning, the counterpart of the reference's debug validator `check_active_blocks`
 * (src/cuda/cuda_sparse_grid.rs:492-650; its call at :479-486 is commented out there): particle ids unique and in range
 * (single-device pipelines), every node block resolved by the hash and the block list Morton-sorted, every particle inside the
 * tile of the particle block that holds it (its associated cell at most one cell outside the 4x4x4 block), all 27 neighbour
 * blocks of a particle block present, block ranges contiguous, disjoint and covering exactly the valid particles.  Re-bins first
 * if the particle set changed.  Every counter is 0 on a healthy pipeline. */
typedef struct {
    uint64_t bad_particle_ids;
    uint64_t blocks_not_in_hash;
    uint64_t particles_outside_their_block;
    uint64_t missing_neighbour_blocks;
    uint64_t coverage_errors;
    uint64_t node_blocks, particle_blocks;   /* sizes checked */
} SparklGridCheck;
int sparkl_hip_grid_check(SparklPipeline* p, SparklGridCheck* out);

/* Counters of the stepping loops since the pipeline was created.  Adaptive step (sparkl_hip_pipeline_step): out[0] substeps that reused the
 * scatter fused into the previous G2P, out[1] substeps that had to scatter again (the predicted step length was wrong),
 * out[2] substeps executed from a look-ahead plan (the rest of the frame enqueued at once while the dt bound repeats),
 * out[3] plans cut short because the bound changed (the substeps behind the change ran as no-ops and were re-planned),
 * out[4] (fixed-dt batches, sparkl_hip_pipeline_substeps) times the deferred part of the velocity gradient was rebuilt
 * because a batch stopped for a re-bin (k_regather_c; SPARKL_HIP_DEFER_C=0 stores it every substep instead),
 * out[5] / out[6] (adaptive step with force_fluids_volume_recomputation) substeps that ran with the volume sweep fused into the
 * substep kernels on a predicted step length, and predictions the estimator refuted (those substeps were repeated in the
 * reference's order; SPARKL_HIP_FLUID_STEP_FUSE=0 turns the prediction off).
 * SPARKL_HIP_LOOKAHEAD=0 in the environment at pipeline creation turns the look-ahead off; =force plans ahead even when
 * the bound has not repeated (test hook for the mismatch path). */
int sparkl_hip_pipeline_step_stats(SparklPipeline* p, uint64_t out[7]);
/* Elapsed GPU time (ms, hipEvents on the pipeline's stream) and launch count of the
 * p2g / grid_update / g2p kernels accumulated since the last reset. */
typedef struct {
    double p2g_ms, grid_update_ms, g2p_ms, sort_ms, dt_ms, halo_ms;
    uint64_t substeps, resorts;
    uint64_t sampled_substeps;   /* substeps whose kernels carried event spans (the *_ms sums cover these) */
    /* fixed-dt runs fuse G2P of substep s with P2G of substep s+1 (k_g2p2g); launches counted per kernel */
    double g2p2g_ms;
    uint64_t p2g_launches, grid_update_launches, g2p_launches, g2p2g_launches;
    double reorder_ms;           /* k_block_reorder launches (most return at once: see sparkl_hip_reorder_policy_set) */
    uint64_t reorder_launches;
    /* force_fluids_volume_recomputation sweep (fluids_volume.rs:9-115): mass-only scatter, mass-only node merge, volume gather */
    double p2g_mass_ms, grid_update_mass_ms, g2p_volume_ms;
    uint64_t p2g_mass_launches, grid_update_mass_launches, g2p_volume_launches;
    /* launches that returned at once behind a raised abort flag (the rest of a batch cut by a re-bin request): spans shorter than
     * a quarter of the longest span of the same kernel; NOT part of the *_ms sums and *_launches counts above */
    double noop_ms;
    uint64_t noop_launches;
} SparklKernelClock;
/* enable: 0 = off, 1 = spans on every substep, N > 1 = spans on every N-th substep */
int sparkl_hip_pipeline_clock_enable(SparklPipeline* p, int enable);
int sparkl_hip_pipeline_clock_read(SparklPipeline* p, SparklKernelClock* out, int reset);

/* Sparse-grid introspection (parity tests): active node blocks after the last substep. */
typedef struct {
    int32_t cell[3];        /* integer node coordinate (position = cell * cell_width) */
    float mass;
    float momentum[3];
    float velocity[3];
} SparklGridNode;
uint64_t sparkl_hip_grid_num_active_blocks(SparklPipeline* p);
uint64_t sparkl_hip_grid_num_particle_blocks(SparklPipeline* p);
/* Writes up to `cap` nodes with mass != 0; returns the count through *n_out.
 * Momentum is only kept when debug capture is on (sparkl_hip_grid_debug_capture). */
int sparkl_hip_grid_debug_capture(SparklPipeline* p, int enable);
int sparkl_hip_grid_read_nodes(SparklPipeline* p, SparklGridNode* out, uint64_t cap, uint64_t* n_out);

/* ---- multi-GPU slabs: one process per GPU, the exchange itself is the host's (RCCL over xGMI) ----
 * A slab owns the particles whose HOME BLOCK index along `axis` is in [block_lo, block_hi)
 * (INT32_MIN / INT32_MAX = open side).  A particle of the last owned block plane writes nodes of the
 * block planes cut-1 and cut, and so does the neighbour's first plane: per substep each side packs its
 * partial (mass, momentum) of the node blocks in those two planes (halo_pack), the host swaps the
 * buffers with the neighbour rank (ncclSend/ncclRecv) and hands the received one back (halo_merge);
 * a + b is bit-identical on both sides, so both run the same grid update and no second exchange is
 * needed.  The re-bin is collective: the per-substep request flag is exported, all-reduced (max) by
 * the host and imported, so every rank skips the same substeps; at a re-bin, particles whose home
 * block left the slab are packed for the neighbour (emigrants) and appended there (immigrants).
 * The reference's scheme for contrast: src/cuda/cuda_sparse_grid.rs:255-322,
 * src/cuda/cuda_mpm_pipeline.rs:407-533 (disabled: testbed_plugin.rs:182-189).  See DESIGN.md. */
typedef struct {
    int32_t axis;           /* 0, 1, 2 */
    int32_t block_lo;       /* first owned block index along axis (INT32_MIN: open) */
    int32_t block_hi;       /* one past the last owned block (INT32_MAX: open) */
    uint32_t _pad0;
    uint64_t id_offset;     /* global id of this rank's first uploaded particle */
} SparklSlab;
int sparkl_hip_slab_set(SparklPipeline* p, const SparklSlab* slab);   /* before particles_upload */
uint64_t sparkl_hip_halo_record_bytes(void);        /* key + 64 x (mass, px, py, pz) */
uint64_t sparkl_hip_migration_record_bytes(void);   /* full particle state */
/* (re-)bin the particles now; afterwards emigrant/halo counts are valid.  Collective by convention. */
int sparkl_hip_rebin(SparklPipeline* p);
int sparkl_hip_slab_emigrant_count(SparklPipeline* p, int side, uint64_t* n);          /* side 0 = low cut, 1 = high */
int sparkl_hip_slab_emigrants_pack(SparklPipeline* p, int side, void* dev_out);        /* removes them locally */
int sparkl_hip_slab_immigrants_add(SparklPipeline* p, const void* dev_in, uint64_t n);  /* needs a re-bin afterwards */
/* Adaptive stepping across slabs: the local dt bound (timestep_estimator.rs:10-51 over this rank's particles, f32 bits,
 * f32::MAX when the rank owns none) written to 4 bytes of device memory; the host takes the MIN over ranks (the bits of a
 * non-negative float order like the float) and applies the step-length rule of mls_solver.rs:33-59.  `sparkl_amd/slab.py`
 * (`SlabStepper.step`) does exactly that. */
int sparkl_hip_dt_bound_export(SparklPipeline* p, void* dev_u32);
/* re-bin request flag of the UPCOMING substep <-> a device u32 the host all-reduces (max) */
int sparkl_hip_flags_export(SparklPipeline* p, void* dev_u32);
int sparkl_hip_flags_import(SparklPipeline* p, const void* dev_u32);
/* split-phase substep: begin = P2G; [halo_pack -> exchange -> halo_merge] per side; finish = grid update + G2P */
int sparkl_hip_substep_begin(SparklPipeline* p, const SparklSolverParams* params, const float gravity[3], float dt);
int sparkl_hip_halo_count(SparklPipeline* p, int side, uint64_t* n);
int sparkl_hip_halo_pack(SparklPipeline* p, int side, void* dev_out);
int sparkl_hip_halo_merge(SparklPipeline* p, int side, const void* dev_in, uint64_t n);
int sparkl_hip_substep_finish(SparklPipeline* p);
/* same, but G2P also scatters the NEXT substep's tiles (k_g2p2g): valid only if the next substep_begin uses the same dt,
 * gravity and parameters and no particle data is touched in between; that begin then launches nothing. */
int sparkl_hip_substep_finish_fused(SparklPipeline* p);
/* synchronises; steps_done counts substeps whose G2P really ran, need_rebin != 0 when a re-bin is due */
int sparkl_hip_substep_status(SparklPipeline* p, uint32_t* steps_done, uint32_t* need_rebin);

/* ---- the exchange inside the library: slab groups (RCCL for bootstrap, re-bins and as the fallback exchange) ----
 * `CudaMpmPipeline::step(.., &mut [SingleGpuMpmContext])` (cuda_mpm_pipeline.rs:262-271) drives every device from one host
 * thread and moves the halo itself (cuda_mpm_pipeline.rs:407-533).  sparkl_hip_group_substeps is that entry point for the slab
 * decomposition: `members` are the slab pipelines THIS process owns, ascending along the slab axis; each has been given its
 * slab (sparkl_hip_slab_set), particles, models and colliders.  Their outer neighbours are reached through RCCL: every member
 * joins the job-wide communicator with sparkl_hip_comm_init(p, id, rank, world) -- rank r owns slab r, its neighbours are
 * ranks r - 1 and r + 1 (one process per GPU: one member per process; one process for all GPUs: one member and one
 * communicator per device).  The 128-byte id comes from sparkl_hip_comm_unique_id on one rank and travels to the others by
 * whatever bootstrap the host has (MPI, a file, torch.distributed).  Per substep the library runs, stream-ordered and without
 * host synchronisation, ONE kernel beyond a single pipeline's two (round 6, "peer-direct"): at its first collective re-bin the job maps
 * every member's exchange arena into every other member -- the pointer itself inside a process, with peer access between its devices as
 * cuda_mpm_pipeline.rs:458-472 enables it between its contexts; hipIpcGetMemHandle / hipIpcOpenMemHandle between the processes of a node --
 * and from then on the re-bin request and the shared node planes are STORED into the neighbours' memory by that kernel and awaited inside
 * the grid update; no collective per substep.  A setup step that fails on any rank (no peer access, hipIpc refused, the handshake through
 * the mapped arenas times out) is agreed collectively and the job runs the round-5 chain instead: MAX all-reduce of the 4-byte re-bin
 * request, halo pack, grouped ncclSend / ncclRecv of the shared planes with both neighbours, halo merge (also: SPARKL_HIP_DIRECT=0, the
 * fluid volume sweep and the eigenerosion psi sweep).  Between the ranks of a communicator the loop times one undisturbed batch on each exchange it has
 * (peer-direct, staged in line, staged overlapped) early in the job and keeps the fastest on every rank: the cost of a remote store is the link's
 * (SPARKL_HIP_DIRECT=1: direct without timing).  Either way the host reads the control block once per batch of 16 substeps and
 * re-bins (with particle migration across the cuts) are collective, through RCCL.
 * Members WITHOUT a communicator must share one device and one stream and exchange by device copies: a single-GPU test mode
 * of the same loop.  Fixed-length substeps only (adaptive steps across ranks: sparkl_hip_dt_bound_export + a MIN all-reduce,
 * see sparkl_amd/slab.py).  librccl is loaded on first use (dlopen); SPARKL_ERR_UNSUPPORTED if it is absent. */
int sparkl_hip_comm_unique_id(void* out_128_bytes);
int sparkl_hip_comm_init(SparklPipeline* p, const void* unique_id_128_bytes, int rank, int world);
/* One host thread driving several devices, as `CudaMpmPipeline::step(.., contexts: &mut [SingleGpuMpmContext])` does
 * (cuda_mpm_pipeline.rs:262-271, make_current :233-237): member i (each on its own device) joins the job as rank first_rank + i; the
 * ncclCommInitRank calls are grouped, so one thread can own all of them (sparkl_hip_comm_init per member from one thread would deadlock). */
int sparkl_hip_comm_init_all(SparklPipeline** members, uint32_t n_members, const void* unique_id_128_bytes, int first_rank, int world);
/* Failure behaviour of the group calls: an error on this rank aborts the communicators of its members (ncclCommAbort), so the peers'
 * pending operations fail instead of hanging; every host wait behind an RCCL operation polls ncclCommGetAsyncError and gives up after
 * SPARKL_HIP_COMM_TIMEOUT_S seconds (default 300, 0 = never).  After such an error the handles hold no communicator (a further group
 * call returns an error until sparkl_hip_comm_init / _init_all gives them a new one). */
int sparkl_hip_comm_allreduce_sum(SparklPipeline* p, long long* value);     /* host-synchronous; no-op without a communicator */
int sparkl_hip_group_substeps(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params,
                              const float gravity[3], float dt, uint32_t nsubsteps);
/* `MpmPipeline::step` across the slabs (adaptive substeps; the dt bound is the MIN over every member and rank, one host read per
 * substep as in the reference); returns the simulated time (params->dt, or the substep length with stop_after_one_substep) */
int sparkl_hip_group_step(SparklPipeline** members, uint32_t n_members, const SparklSolverParams* params,
                          const float gravity[3], float* simulated_time, uint32_t* num_substeps_or_null);
/* Halo / compute overlap of the group loop (the reference runs its halo work on a second stream, cuda_mpm_pipeline.rs:410-430).  Between
 * RCCL ranks the schedule is chosen by measurement: the first two undisturbed 16-substep batches run in line and overlapped, the MAX over
 * the ranks of each wall clock is taken (one all-reduce) and every rank keeps the faster one.  SPARKL_HIP_OVERLAP=0/1 forces a schedule;
 * members of one process on one device run in line.  *chosen: 1 overlapped, 0 in line, -1 not decided yet. */
int sparkl_hip_overlap_info(const SparklPipeline* p, int* chosen, double* us_per_substep_2);
uint64_t sparkl_hip_group_rebins(const SparklPipeline* p);                  /* collective re-bins this member has taken part in */
/* Load re-balancing (on by default): at a collective re-bin the two ranks of a cut compare particle counts and occupied block
 * planes and move the cut by one block plane towards the fuller side when that still leaves it the fuller one; the particles of
 * the moved plane migrate like any emigrant.  sparkl_hip_slab_get returns the current range and how often a cut of this member moved. */
int sparkl_hip_slab_balance(SparklPipeline* p, int enable);
int sparkl_hip_slab_get(const SparklPipeline* p, SparklSlab* out, uint64_t* cut_moves_or_null);
/* the hipStream_t the pipeline runs on (to create a second member on the same stream, or to order foreign work against it) */
void* sparkl_hip_pipeline_stream(const SparklPipeline* p);

const char* sparkl_hip_last_error_string(void);
/* "sparkl_hip <version> gfx950" */
const char* sparkl_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPARKL_HIP_H */
