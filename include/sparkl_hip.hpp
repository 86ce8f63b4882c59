// sparkl_hip.hpp -- header-only C++ host side above the C ABI (include/sparkl_hip.h).
//
// The reference's host code is Rust; there is no Rust toolchain in this environment, so the host-language mirror of
// the reference's interface is C++ (same names, argument meaning and error behaviour):
//   Particle::new                    src/dynamics/particle.rs:54-103
//   ParticleSet                      src/dynamics/particle_set.rs:20-73
//   ParticleModel / ParticleModelSet src/dynamics/particle_model.rs:14-18,85-139
//   CorotatedLinearElasticity::new ... DruckerPragerPlasticity::new   src_core/dynamics/models/*.rs
//   SolverParameters::default        src_core/dynamics/solver/solver_parameters.rs:56-68
//   extract_particles_components     src/cuda/cuda_particle_set.rs:189-248
//   MpmPipeline::step                src/pipelines/mpm_pipeline.rs:15-42  /  CudaMpmPipeline::step  src/cuda/cuda_mpm_pipeline.rs:262-271
// Errors: where the reference returns `CudaResult` / panics, these throw `sparkl::Error` carrying the C ABI code.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sparkl_hip.h"

namespace sparkl {

using Real = float;
using Vector = std::array<Real, 3>;
using Matrix = std::array<Real, 9>;  // column-major, like nalgebra::Matrix3

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error("sparkl_hip error " + std::to_string(c) + ": " + what), code(c) {}
};
inline void check(int code) {
    if (code != SPARKL_OK) throw Error(code, sparkl_hip_last_error_string());
}

// ---- material models (constructors compute Lame parameters in f32 exactly like src_core/utils/physics.rs:18-27)
inline std::pair<Real, Real> lame_lambda_mu(Real young_modulus, Real poisson_ratio) {
    return {young_modulus * poisson_ratio / ((1.0f + poisson_ratio) * (1.0f - 2.0f * poisson_ratio)), young_modulus / (2.0f * (1.0f + poisson_ratio))};
}
struct CorotatedLinearElasticity {
    bool split_stress_on_failure = true;
    Real cfl_coeff = 0.9f, lambda = 0, mu = 0;
    CorotatedLinearElasticity(Real young_modulus, Real poisson_ratio) { std::tie(lambda, mu) = lame_lambda_mu(young_modulus, poisson_ratio); }
    bool is_fluid() const { return false; }
    SparklConstitutiveModel to_core_model() const {
        SparklConstitutiveModel m{};
        m.tag = SPARKL_CONSTITUTIVE_COROTATED_LINEAR;
        m.u.corotated.split_stress_on_failure = split_stress_on_failure;
        m.u.corotated.cfl_coeff = cfl_coeff; m.u.corotated.lambda = lambda; m.u.corotated.mu = mu;
        return m;
    }
};
struct NeoHookeanElasticity {
    Real cfl_coeff = 0.5f, lambda = 0, mu = 0;
    NeoHookeanElasticity(Real young_modulus, Real poisson_ratio) { std::tie(lambda, mu) = lame_lambda_mu(young_modulus, poisson_ratio); }
    bool is_fluid() const { return false; }
    SparklConstitutiveModel to_core_model() const {
        SparklConstitutiveModel m{};
        m.tag = SPARKL_CONSTITUTIVE_NEO_HOOKEAN;
        m.u.neo_hookean.cfl_coeff = cfl_coeff; m.u.neo_hookean.lambda = lambda; m.u.neo_hookean.mu = mu;
        return m;
    }
};
struct MonaghanSphEos {
    Real pressure0; uint32_t gamma; Real viscosity; Real max_neg_pressure = 1.0f;
    MonaghanSphEos(Real p0, uint32_t g, Real visc) : pressure0(p0), gamma(g), viscosity(visc) {}
    bool is_fluid() const { return true; }
    SparklConstitutiveModel to_core_model() const {
        SparklConstitutiveModel m{};
        m.tag = SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH;
        m.u.monaghan.pressure0 = pressure0; m.u.monaghan.gamma = gamma; m.u.monaghan.viscosity = viscosity; m.u.monaghan.max_neg_pressure = max_neg_pressure;
        return m;
    }
};
struct DruckerPragerPlasticity {
    Real h0, h1, h2 = 0.2f, h3, lambda = 0, mu = 0;
    bool only_active_when_failed = false;
    Real volume_correction = 1.0f;
    DruckerPragerPlasticity(Real young_modulus, Real poisson_ratio) {
        const Real d2r = 3.14159265358979323846f / 180.0f;
        h0 = 35.0f * d2r; h1 = 9.0f * d2r; h3 = 10.0f * d2r;
        std::tie(lambda, mu) = lame_lambda_mu(young_modulus, poisson_ratio);
    }
    SparklPlasticModel to_core_model() const {
        SparklPlasticModel m{};
        m.tag = SPARKL_PLASTIC_DRUCKER_PRAGER;
        auto& d = m.u.drucker_prager;
        d.h0 = h0; d.h1 = h1; d.h2 = h2; d.h3 = h3; d.lambda = lambda; d.mu = mu;
        d.only_active_when_failed = only_active_when_failed; d.volume_correction = volume_correction;
        return m;
    }
};

struct SnowPlasticity {                     // plasticity_snow.rs:18-24
    Real min_epsilon = 2.5e-2f, max_epsilon = 4.5e-3f, hardening_coeff = 10.0f;
    SparklPlasticModel to_core_model() const {
        SparklPlasticModel m{};
        m.tag = SPARKL_PLASTIC_SNOW;
        m.u.snow.min_epsilon = min_epsilon; m.u.snow.max_epsilon = max_epsilon; m.u.snow.hardening_coeff = hardening_coeff;
        return m;
    }
};
struct RankinePlasticity {                  // plasticity_rankine.rs:19-32
    Real mu = 0, lambda = 0, tensile_strength, softening_rate;
    RankinePlasticity(Real young_modulus, Real poisson_ratio, Real tensile_strength_, Real softening_rate_)
        : tensile_strength(tensile_strength_), softening_rate(softening_rate_) { std::tie(lambda, mu) = lame_lambda_mu(young_modulus, poisson_ratio); }
    SparklPlasticModel to_core_model() const {
        SparklPlasticModel m{};
        m.tag = SPARKL_PLASTIC_RANKINE;
        m.u.rankine.mu = mu; m.u.rankine.lambda = lambda; m.u.rankine.tensile_strength = tensile_strength; m.u.rankine.softening_rate = softening_rate;
        return m;
    }
};
struct NaccPlasticity {                     // plasticity_nacc.rs:22-43 (dim3: d = 3)
    Real mu, kappa, hardening_factor, cohesion, friction;
    bool hardening_enabled;
    NaccPlasticity(Real young_modulus, Real poisson_ratio, Real cohesion_, bool hardening_enabled_, Real hardening_factor_, Real friction_angle)
        : mu(young_modulus / (2.0f * (1.0f + poisson_ratio))), kappa(young_modulus / (3.0f * (1.0f - 2.0f * poisson_ratio))),
          hardening_factor(hardening_factor_), cohesion(cohesion_), hardening_enabled(hardening_enabled_) {
        const Real sin_f = std::sin(friction_angle), d = 3.0f;
        friction = std::sqrt(2.0f / 3.0f) * 2.0f * sin_f / (3.0f - sin_f) * d / std::sqrt(2.0f / (6.0f - d));
    }
    SparklPlasticModel to_core_model() const {
        SparklPlasticModel m{};
        m.tag = SPARKL_PLASTIC_NACC;
        auto& n = m.u.nacc;
        n.mu = mu; n.kappa = kappa; n.hardening_enabled = hardening_enabled; n.hardening_factor = hardening_factor; n.cohesion = cohesion; n.friction = friction;
        return m;
    }
};
struct MaximumStressFailure {               // failure_maximum_stress.rs:13-18
    Real max_principal_stress, max_shear_stress;
    SparklFailureModel to_core_model() const {
        SparklFailureModel m{};
        m.tag = SPARKL_FAILURE_MAXIMUM_STRESS;
        m.u.maximum_stress.max_principal_stress = max_principal_stress; m.u.maximum_stress.max_shear_stress = max_shear_stress;
        return m;
    }
};

using ParticleModelHandle = uint64_t;
struct ParticleModel {
    SparklParticleModel core{};
    template <class C> static ParticleModel new_(const C& constitutive) {  // ParticleModel::new
        ParticleModel m;
        m.core.constitutive_model = constitutive.to_core_model();
        return m;
    }
    template <class C, class P> static ParticleModel with_plasticity(const C& constitutive, const P& plasticity) {
        ParticleModel m = new_(constitutive);
        m.core.has_plastic_model = 1;
        m.core.plastic_model = plasticity.to_core_model();
        return m;
    }
    template <class C, class Fm> static ParticleModel with_failure(const C& constitutive, const Fm& failure) {  // particle_model.rs:104-113
        ParticleModel m = new_(constitutive);
        m.core.has_failure_model = 1;
        m.core.failure_model = failure.to_core_model();
        return m;
    }
};
class ParticleModelSet {
    std::vector<ParticleModel> models_;
public:
    ParticleModelHandle insert(const ParticleModel& m) { models_.push_back(m); return models_.size() - 1; }
    const ParticleModel& get(ParticleModelHandle h) const { return models_.at(h); }
    size_t len() const { return models_.size(); }
    std::vector<SparklParticleModel> to_core() const {
        std::vector<SparklParticleModel> v;
        for (auto& m : models_) v.push_back(m.core);
        return v;
    }
};

// ---- particles (AoS on the host, like the reference)
struct Particle {
    ParticleModelHandle model = 0;
    Real mass = 0, volume0 = 0, radius0 = 0;
    Vector position{}, velocity{};
    std::optional<Vector> kinematic_vel;
    Matrix velocity_gradient{}, deformation_gradient{1, 0, 0, 0, 1, 0, 0, 0, 1};
    Real plastic_deformation_gradient_det = 1.0f;
    bool failed = false, is_static = false;
    Real psi_pos = 0, phase = 1.0f, nacc_alpha = -0.01f, plastic_hardening = 1.0f, elastic_hardening = 1.0f, log_vol_gain = 0;

    static Particle new_(ParticleModelHandle model, Vector position, Real radius, Real density0) {  // Particle::new
        Particle p;
        p.model = model; p.position = position; p.radius0 = radius;
        const Real d = radius * 2.0f;
        p.volume0 = d * d * d;
        p.mass = p.volume0 * density0;
        return p;
    }
};
class ParticleSet {
    std::vector<Particle> particles_;
public:
    size_t len() const { return particles_.size(); }
    const std::vector<Particle>& particles() const { return particles_; }
    std::vector<Particle>& particles_mut() { return particles_; }
    void insert(const Particle& p) { particles_.push_back(p); }
    void insert_batch(const std::vector<Particle>& ps) { particles_.insert(particles_.end(), ps.begin(), ps.end()); }
    void remove_range(size_t start, size_t end) { particles_.erase(particles_.begin() + start, particles_.begin() + end); }
};
// examples3d/helper.rs:6-40 (randomize = false)
inline std::vector<Particle> cube_particles(Vector origin, size_t ni, size_t nj, size_t nk, ParticleModelHandle model, Real particle_rad, Real density) {
    std::vector<Particle> out;
    out.reserve(ni * nj * nk);
    for (size_t i = 0; i < ni; ++i)
        for (size_t j = 0; j < nj; ++j)
            for (size_t k = 0; k < nk; ++k)
                out.push_back(Particle::new_(model, {origin[0] + (Real)i * particle_rad * 2.0f, origin[1] + (Real)j * particle_rad * 2.0f, origin[2] + (Real)k * particle_rad * 2.0f},
                                             particle_rad, density));
    return out;
}

struct ParticleComponents {
    std::vector<SparklParticlePosition> position;
    std::vector<SparklParticleVelocity> velocity;
    std::vector<SparklParticleVolume> volume;
    std::vector<SparklParticleStatus> status;
    std::vector<SparklParticlePhase> phase;
    std::vector<SparklParticleData> data;
    std::vector<SparklVelocityGradient> velocity_gradient;
};
inline ParticleComponents extract_particles_components(const std::vector<Particle>& ps) {
    ParticleComponents c;
    const size_t n = ps.size();
    c.position.resize(n); c.velocity.resize(n); c.volume.resize(n); c.status.resize(n); c.phase.resize(n); c.data.resize(n); c.velocity_gradient.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const Particle& p = ps[i];
        std::memcpy(c.position[i].point, p.position.data(), 12);
        std::memcpy(c.velocity[i].vector, p.velocity.data(), 12);
        c.volume[i] = SparklParticleVolume{p.mass, p.volume0, p.radius0, {}, p.plastic_deformation_gradient_det};
        std::memcpy(c.volume[i].deformation_gradient, p.deformation_gradient.data(), 36);
        std::memset(&c.status[i], 0, sizeof(SparklParticleStatus));
        c.status[i].failed = p.failed; c.status[i].is_static = p.is_static; c.status[i].kinematic_vel_enabled = p.kinematic_vel.has_value();
        if (p.kinematic_vel) std::memcpy(c.status[i].kinematic_vel, p.kinematic_vel->data(), 12);
        c.status[i].model_index = p.model;
        c.phase[i] = SparklParticlePhase{p.phase, p.psi_pos};
        c.data[i] = SparklParticleData{p.nacc_alpha, p.plastic_hardening, p.elastic_hardening, p.log_vol_gain};
        std::memcpy(c.velocity_gradient[i].m, p.velocity_gradient.data(), 36);
    }
    return c;
}

struct SolverParameters : SparklSolverParams {
    SolverParameters() {  // impl Default
        dt = 1.0f / 60.0f; max_substep_dt = std::numeric_limits<float>::max(); max_num_substeps = 1000;
        boundary_handling = SPARKL_BOUNDARY_FRICTION; damage_model = SPARKL_DAMAGE_NONE;
        force_fluids_volume_recomputation = 0; enable_boundary_particle_projection = 0; stop_after_one_substep = 0; _pad0 = 0;
        simulation_dofs = SPARKL_DOFS_LOCK_NONE;
    }
};

// ---- the pipeline: MpmPipeline::step over a device-resident particle set (CudaParticleSet + CudaMpmPipeline rolled into one handle)
class MpmPipeline {
    SparklPipeline* h_ = nullptr;
public:
    explicit MpmPipeline(Real cell_width, int device_id = 0, uint32_t domain_log2_bytes = 30, void* stream = nullptr) {
        SparklPipelineDesc d{device_id, cell_width, domain_log2_bytes, stream};
        check(sparkl_hip_pipeline_create(&d, &h_));
    }
    ~MpmPipeline() { sparkl_hip_pipeline_destroy(h_); }
    MpmPipeline(const MpmPipeline&) = delete;
    MpmPipeline& operator=(const MpmPipeline&) = delete;
    SparklPipeline* raw() { return h_; }

    void set_models(const ParticleModelSet& models) {  // CudaParticleModelSet::from_model_set
        auto core = models.to_core();
        check(sparkl_hip_models_set(h_, (uint32_t)core.size(), core.data()));
    }
    void set_colliders(const std::vector<SparklCuboidCollider>& cuboids) { check(sparkl_hip_colliders_set_cuboids(h_, (uint32_t)cuboids.size(), cuboids.data())); }
    // heightfield colliders (rapier `ColliderBuilder::heightfield`); heights are copied, the pointers need not outlive the call
    void set_heightfields(const std::vector<SparklHeightFieldCollider>& hfs) { check(sparkl_hip_colliders_set_heightfields(h_, (uint32_t)hfs.size(), hfs.data())); }
    // triangle meshes (rapier `ColliderBuilder::trimesh`); the mesh is copied
    void set_trimeshes(const std::vector<SparklTriMeshCollider>& meshes) { check(sparkl_hip_colliders_set_trimeshes(h_, (uint32_t)meshes.size(), meshes.data())); }
    // timings.substeps of the last step() that asked for timings (cuda_mpm_pipeline.rs:28-85)
    std::vector<SparklSubstepTimings> step_substep_timings() {
        uint32_t n = 0;
        check(sparkl_hip_pipeline_step_substep_timings(h_, nullptr, 0, &n));
        std::vector<SparklSubstepTimings> out(n);
        if (n) check(sparkl_hip_pipeline_step_substep_timings(h_, out.data(), n, &n));
        return out;
    }
    // CudaColliderOptions per collider kind (cuda_collider_set.rs:22-58), in the order of the matching set_* call
    void set_cuboid_options(const std::vector<SparklColliderOptions>& o) { check(sparkl_hip_colliders_set_cuboid_options(h_, (uint32_t)o.size(), o.data())); }
    void set_heightfield_options(const std::vector<SparklColliderOptions>& o) { check(sparkl_hip_colliders_set_heightfield_options(h_, (uint32_t)o.size(), o.data())); }
    void set_polylines(const std::vector<SparklPolylineCollider>& pl) { check(sparkl_hip_colliders_set_polylines(h_, (uint32_t)pl.size(), pl.data())); }
    void set_polyline_options(const std::vector<SparklColliderOptions>& o) { check(sparkl_hip_colliders_set_polyline_options(h_, (uint32_t)o.size(), o.data())); }
    void enable_penalty_force(bool on) { check(sparkl_hip_colliders_enable_penalty_force(h_, on ? 1 : 0)); }   // off = the reference (`if false`)
    void set_trimesh_options(const std::vector<SparklColliderOptions>& o) { check(sparkl_hip_colliders_set_trimesh_options(h_, (uint32_t)o.size(), o.data())); }
    // per-frame render readback (testbed_plugin.rs:533-588): pinned buffers, copies overlapped with the next step
    template <class T> static T* host_alloc(size_t count) { void* ptr = nullptr; check(sparkl_hip_host_alloc(&ptr, count * sizeof(T))); return (T*)ptr; }
    static void host_free(void* ptr) { check(sparkl_hip_host_free(ptr)); }
    void readback_begin(SparklParticlePosition* pos, SparklParticleVelocity* vel, SparklParticlePhase* phase) { check(sparkl_hip_readback_begin(h_, pos, vel, phase, len())); }
    void readback_wait() { check(sparkl_hip_readback_wait(h_)); }
    void upload(const ParticleSet& particles) {  // CudaParticleSet::from_particles(extract_particles_components(..))
        ParticleComponents c = extract_particles_components(particles.particles());
        check(sparkl_hip_particles_upload(h_, c.status.size(), c.status.data(), c.position.data(), c.velocity.data(), c.volume.data(), c.phase.data(), c.data.data(),
                                          c.velocity_gradient.data()));
    }
    size_t len() const { return (size_t)sparkl_hip_particles_len(h_); }
    // advance by params.dt; returns the simulated time like MpmPipeline::step
    Real step(const SolverParameters& params, const Vector& gravity, SparklTimings* timings = nullptr) {
        float t = 0;
        check(sparkl_hip_pipeline_step(h_, &params, gravity.data(), timings, &t));
        return t;
    }
    void substeps(const SolverParameters& params, const Vector& gravity, Real dt, uint32_t n) { check(sparkl_hip_pipeline_substeps(h_, &params, gravity.data(), dt, n)); }
    // ---- slabs (multi-GPU): `CudaMpmPipeline::step(.., &mut [SingleGpuMpmContext])` for a spatial decomposition.  Give every slab
    // its block range before uploading its particles, join the RCCL communicator (one rank per slab) or keep all slabs on one
    // device and stream, then step them together with group_substeps.
    void* stream() const { return sparkl_hip_pipeline_stream(h_); }
    void set_slab(int axis, int32_t block_lo, int32_t block_hi, uint64_t id_offset) {
        SparklSlab s{axis, block_lo, block_hi, 0u, id_offset};
        check(sparkl_hip_slab_set(h_, &s));
    }
    void comm_init(const void* unique_id_128_bytes, int rank, int world) { check(sparkl_hip_comm_init(h_, unique_id_128_bytes, rank, world)); }
    // one host thread, one member per device (cuda_mpm_pipeline.rs:262-271): member i becomes rank first_rank + i
    static void comm_init_all(const std::vector<MpmPipeline*>& members, const void* unique_id_128_bytes, int first_rank, int world) {
        std::vector<SparklPipeline*> raw;
        for (MpmPipeline* m : members) raw.push_back(m->h_);
        check(sparkl_hip_comm_init_all(raw.data(), (uint32_t)raw.size(), unique_id_128_bytes, first_rank, world));
    }
    static void group_substeps(const std::vector<MpmPipeline*>& members, const SolverParameters& params, const Vector& gravity, Real dt, uint32_t n) {
        std::vector<SparklPipeline*> raw;
        for (auto* m : members) raw.push_back(m->h_);
        check(sparkl_hip_group_substeps(raw.data(), (uint32_t)raw.size(), &params, gravity.data(), dt, n));
    }
    // `check_active_blocks` of the reference (cuda_sparse_grid.rs:492-650): every counter of the result is 0 on a healthy pipeline
    SparklGridCheck grid_check() { SparklGridCheck c; check(sparkl_hip_grid_check(h_, &c)); return c; }
    bool grid_is_consistent() {
        const SparklGridCheck c = grid_check();
        return !(c.bad_particle_ids | c.blocks_not_in_hash | c.particles_outside_their_block | c.missing_neighbour_blocks | c.coverage_errors);
    }
    // device order of a slab pipeline: positions / velocities as stored, plus the global id of every record
    void read_device_order(std::vector<SparklParticlePosition>& pos, std::vector<SparklParticleVelocity>& vel, std::vector<uint32_t>& ids) {
        const size_t n = len();
        pos.resize(n); vel.resize(n); ids.resize(n);
        check(sparkl_hip_particles_read_positions(h_, pos.data(), n)); check(sparkl_hip_particles_read_velocities(h_, vel.data(), n));
        check(sparkl_hip_particles_read_sorted_ids(h_, ids.data(), n));
    }
    // testbed_plugin.rs:533-588: read positions / velocities / F back into the CPU ParticleSet
    void download(ParticleSet& particles) {
        const size_t n = len();
        if (n != particles.len()) throw Error(SPARKL_ERR_INVALID_ARGUMENT, "particle count changed");
        std::vector<SparklParticlePosition> pos(n); std::vector<SparklParticleVelocity> vel(n); std::vector<SparklParticleVolume> vol(n);
        std::vector<SparklParticleStatus> st(n); std::vector<SparklParticlePhase> ph(n); std::vector<SparklParticleData> da(n); std::vector<SparklVelocityGradient> vg(n);
        check(sparkl_hip_particles_read_positions(h_, pos.data(), n)); check(sparkl_hip_particles_read_velocities(h_, vel.data(), n));
        check(sparkl_hip_particles_read_volumes(h_, vol.data(), n)); check(sparkl_hip_particles_read_statuses(h_, st.data(), n));
        check(sparkl_hip_particles_read_phases(h_, ph.data(), n)); check(sparkl_hip_particles_read_data(h_, da.data(), n));
        check(sparkl_hip_particles_read_velocity_gradients(h_, vg.data(), n));
        auto& ps = particles.particles_mut();
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(ps[i].position.data(), pos[i].point, 12); std::memcpy(ps[i].velocity.data(), vel[i].vector, 12);
            std::memcpy(ps[i].deformation_gradient.data(), vol[i].deformation_gradient, 36); std::memcpy(ps[i].velocity_gradient.data(), vg[i].m, 36);
            ps[i].plastic_deformation_gradient_det = vol[i].plastic_deformation_gradient_det;
            ps[i].failed = st[i].failed; ps[i].psi_pos = ph[i].psi_pos; ps[i].phase = ph[i].phase;
            ps[i].plastic_hardening = da[i].plastic_hardening; ps[i].log_vol_gain = da[i].log_vol_gain;
            ps[i].elastic_hardening = da[i].elastic_hardening; ps[i].nacc_alpha = da[i].nacc_alpha;   // snow / NACC rewrite them
        }
    }
};

}  // namespace sparkl
