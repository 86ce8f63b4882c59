// ORACLE -- TEST INFRASTRUCTURE ONLY (see mpm_oracle.hpp).  extern "C" surface for ctypes.
#include "mpm_oracle.hpp"

#include <array>
#include <chrono>

using namespace oracle;

namespace {
struct Handle {
    bool f64;
    Oracle<float>* of = nullptr;
    Oracle<double>* od = nullptr;
};

template <class R>
void set_particles(Oracle<R>& o, uint64_t n, const SparklParticleStatus* st, const SparklParticlePosition* pos,
                   const SparklParticleVelocity* vel, const SparklParticleVolume* vol, const SparklParticlePhase* ph,
                   const SparklParticleData* data, const SparklVelocityGradient* vg) {
    o.particles.resize(n);
    o.order.resize(n);   // ParticleSet::set_particles: order = (0..n).collect() (particle_set.rs:48-51)
    // parallel fill = first touch: the pages of the particle vector land on the NUMA nodes of the threads that fill them, not all on the
    // loading thread's node (o.particles does not value-initialise: no_init_allocator)
#pragma omp parallel for schedule(static) num_threads(o.num_threads)
    for (uint64_t i = 0; i < n; ++i) o.order[i] = i;
#pragma omp parallel for schedule(static) num_threads(o.num_threads)
    for (uint64_t i = 0; i < n; ++i) {
        Particle<R>& p = o.particles[i];
        p.model = st[i].model_index;
        p.mass = vol[i].mass; p.volume0 = vol[i].volume0; p.radius0 = vol[i].radius0;
        for (int a = 0; a < 3; ++a) { p.position[a] = pos[i].point[a]; p.velocity[a] = vel[i].vector[a]; p.kinematic_vel[a] = st[i].kinematic_vel[a]; }
        p.kinematic_vel_enabled = st[i].kinematic_vel_enabled != 0;
        for (int k = 0; k < 9; ++k) { p.deformation_gradient.m[k] = vol[i].deformation_gradient[k]; p.velocity_gradient.m[k] = vg ? (R)vg[i].m[k] : R(0); }
        p.plastic_deformation_gradient_det = vol[i].plastic_deformation_gradient_det;
        p.grid_index = 0;
        p.failed = st[i].failed != 0; p.is_static = st[i].is_static != 0;
        p.psi_pos = ph[i].psi_pos; p.phase = ph[i].phase;
        p.parameter1 = p.parameter2 = 0; p.crack_propagation_factor = 0; p.crack_threshold = std::numeric_limits<R>::max();
        p.nacc_alpha = data[i].nacc_alpha; p.plastic_hardening = data[i].plastic_hardening;
        p.elastic_hardening = data[i].elastic_hardening; p.log_vol_gain = data[i].log_vol_gain;
    }
}

template <class R>
void get_particles(const Oracle<R>& o, SparklParticleStatus* st, SparklParticlePosition* pos, SparklParticleVelocity* vel,
                   SparklParticleVolume* vol, SparklParticlePhase* ph, SparklParticleData* data, SparklVelocityGradient* vg) {
    for (size_t i = 0; i < o.particles.size(); ++i) {
        const Particle<R>& p = o.particles[i];
        if (st) {
            std::memset(&st[i], 0, sizeof(st[i]));
            st[i].failed = p.failed; st[i].is_static = p.is_static; st[i].kinematic_vel_enabled = p.kinematic_vel_enabled;
            for (int a = 0; a < 3; ++a) st[i].kinematic_vel[a] = (float)p.kinematic_vel[a];
            st[i].model_index = p.model;
        }
        if (pos) for (int a = 0; a < 3; ++a) pos[i].point[a] = (float)p.position[a];
        if (vel) for (int a = 0; a < 3; ++a) vel[i].vector[a] = (float)p.velocity[a];
        if (vol) {
            vol[i].mass = (float)p.mass; vol[i].volume0 = (float)p.volume0; vol[i].radius0 = (float)p.radius0;
            for (int k = 0; k < 9; ++k) vol[i].deformation_gradient[k] = (float)p.deformation_gradient.m[k];
            vol[i].plastic_deformation_gradient_det = (float)p.plastic_deformation_gradient_det;
        }
        if (ph) { ph[i].phase = (float)p.phase; ph[i].psi_pos = (float)p.psi_pos; }
        if (data) {
            data[i].nacc_alpha = (float)p.nacc_alpha; data[i].plastic_hardening = (float)p.plastic_hardening;
            data[i].elastic_hardening = (float)p.elastic_hardening; data[i].log_vol_gain = (float)p.log_vol_gain;
        }
        if (vg) for (int k = 0; k < 9; ++k) vg[i].m[k] = (float)p.velocity_gradient.m[k];
    }
}

template <class R>
void get_state_f64(const Oracle<R>& o, double* pos, double* vel, double* F, double* C) {
    for (size_t i = 0; i < o.particles.size(); ++i) {
        const Particle<R>& p = o.particles[i];
        for (int a = 0; a < 3; ++a) { if (pos) pos[3 * i + a] = (double)p.position[a]; if (vel) vel[3 * i + a] = (double)p.velocity[a]; }
        for (int k = 0; k < 9; ++k) { if (F) F[9 * i + k] = (double)p.deformation_gradient.m[k]; if (C) C[9 * i + k] = (double)p.velocity_gradient.m[k]; }
    }
}

template <class R>
void set_cuboids(Oracle<R>& o, uint32_t n, const SparklCuboidCollider* c) {
    o.cuboids.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) { o.cuboids[i].he[a] = c[i].half_extents[a]; o.cuboids[i].t[a] = c[i].translation[a]; }
        for (int a = 0; a < 4; ++a) o.cuboids[i].q[a] = c[i].rotation[a];
        o.cuboids[i].friction = c[i].friction;
    }
}

template <class R>
void set_heightfields(Oracle<R>& o, uint32_t n, const SparklHeightFieldCollider* c) {
    o.heightfields.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        auto& h = o.heightfields[i];
        h.nrows = (int)c[i].nrows; h.ncols = (int)c[i].ncols;
        h.heights.assign(c[i].heights, c[i].heights + (size_t)c[i].nrows * c[i].ncols);
        for (int a = 0; a < 3; ++a) { h.scale[a] = c[i].scale[a]; h.t[a] = c[i].translation[a]; }
        for (int a = 0; a < 4; ++a) h.q[a] = c[i].rotation[a];
        h.friction = c[i].friction; h.below_is_solid = c[i].below_is_solid != 0;
    }
}

template <class R>
void fill_trimesh(TriMesh<R>& m, const SparklTriMeshCollider& c) {
    m.v.resize(c.num_vertices);
    for (uint32_t k = 0; k < c.num_vertices; ++k) m.v[k] = Vec3<R>{{(R)c.vertices[3 * k], (R)c.vertices[3 * k + 1], (R)c.vertices[3 * k + 2]}};
    m.idx.resize(c.num_triangles);
    for (uint32_t k = 0; k < c.num_triangles; ++k) m.idx[k] = {{c.indices[3 * k], c.indices[3 * k + 1], c.indices[3 * k + 2]}};
    for (int a = 0; a < 3; ++a) m.t[a] = c.translation[a];
    for (int a = 0; a < 4; ++a) m.q[a] = c.rotation[a];
    m.friction = c.friction; m.oriented = c.oriented != 0; m.flip_interior = c.flip_interior != 0;
    if (m.oriented) m.compute_pseudo_normals();
}
template <class R>
void set_polylines(Oracle<R>& o, uint32_t n, const SparklPolylineCollider* c) {
    o.polylines.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        auto& pl = o.polylines[i];
        pl.v.resize(c[i].num_vertices);
        for (uint32_t k = 0; k < c[i].num_vertices; ++k) pl.v[k] = Vec3<R>{{(R)c[i].vertices[3 * k], (R)c[i].vertices[3 * k + 1], (R)c[i].vertices[3 * k + 2]}};
        for (int a = 0; a < 4; ++a) pl.q[a] = c[i].rotation[a];
        for (int a = 0; a < 3; ++a) pl.t[a] = c[i].translation[a];
        pl.friction = c[i].friction;
        pl.opts = ColliderOpts{};
    }
}
template <class R>
void set_trimeshes(Oracle<R>& o, uint32_t n, const SparklTriMeshCollider* c) {
    o.trimeshes.resize(n);
    for (uint32_t i = 0; i < n; ++i) fill_trimesh(o.trimeshes[i], c[i]);
}

template <class R>
uint64_t get_nodes(Oracle<R>& o, SparklGridNode* out, uint64_t cap) {
    uint64_t n = 0;
    for (const auto& reg : o.regions)
        for (uint64_t i = 0; i < 64; ++i) {
            uint64_t id = reg.id | (i << PACK_ALIGN);
            const GridNode<R>& g = o.node(id);
            if (g.mass == R(0)) continue;
            if (n < cap && out) {
                uint32_t c[3]; unpack(id, c);
                for (int a = 0; a < 3; ++a) { out[n].cell[a] = (int32_t)c[a]; out[n].momentum[a] = (float)g.momentum[a]; out[n].velocity[a] = (float)g.velocity[a]; }
                out[n].mass = (float)g.mass;
            }
            ++n;
        }
    return n;
}
// generic plastic update: state = {Jp, plastic_hardening, log_vol_gain, elastic_hardening, nacc_alpha}
template <class R> void plastic_update_t(const SparklPlasticModel* pm, double phase, double F[9], double st[5]) {
    Mat3<R> f; for (int i = 0; i < 9; ++i) f.m[i] = (R)F[i];
    R jp = (R)st[0], q = (R)st[1], lvg = (R)st[2], eh = (R)st[3], al = (R)st[4];
    switch (pm->tag) {
        case SPARKL_PLASTIC_DRUCKER_PRAGER: dp_update_particle<R>(pm->u.drucker_prager, (R)phase, f, jp, q, lvg); break;
        case SPARKL_PLASTIC_SNOW: snow_update_particle<R>(pm->u.snow, f, eh, jp); break;
        case SPARKL_PLASTIC_RANKINE: rankine_update_particle<R>(pm->u.rankine, f, q); break;
        case SPARKL_PLASTIC_NACC: nacc_update_particle<R>(pm->u.nacc, f, al); break;
        default: break;
    }
    for (int i = 0; i < 9; ++i) F[i] = f.m[i];
    st[0] = jp; st[1] = q; st[2] = lvg; st[3] = eh; st[4] = al;
}
template <class R> static void stress_impl(const SparklConstitutiveModel* cm, double phase, double hardening, const double* F, double mass,
                                          double volume0, double fluid_j, const double* C, double* out) {
    Mat3<R> f, c; for (int i = 0; i < 9; ++i) { f.m[i] = (R)F[i]; c.m[i] = (R)C[i]; }
    Mat3<R> s = Mat3<R>::zeros();
    switch (cm->tag) {
        case SPARKL_CONSTITUTIVE_COROTATED_LINEAR: s = corotated_kirchhoff<R>(cm->u.corotated, (R)phase, (R)hardening, f); break;
        case SPARKL_CONSTITUTIVE_NEO_HOOKEAN: s = neo_hookean_kirchhoff<R>(cm->u.neo_hookean, (R)phase, (R)hardening, f); break;
        case SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH: {
            R density_fluid = ((R)mass / (R)volume0) / f(0, 0);
            s = monaghan_kirchhoff<R>(cm->u.monaghan, (R)mass, (R)volume0, density_fluid, (R)fluid_j, c);
            break;
        }
    }
    for (int i = 0; i < 9; ++i) out[i] = (double)s.m[i];
}
template <class R> static bool region_active(Oracle<R>& o, uint64_t id) { for (auto& r : o.regions) if (r.id == id) return true; return false; }
}  // namespace

#define DISPATCH(h, expr_f, expr_d) (((Handle*)(h))->f64 ? (expr_d) : (expr_f))

// SparklColliderOptions per collider kind (0 cuboids, 1 heightfields, 2 trimeshes), in the order of the matching oracle_set_* call
template <class R>
void set_options(Oracle<R>& o, int kind, uint32_t n, const SparklColliderOptions* opt) {
    auto put = [&](ColliderOpts& dst, const SparklColliderOptions& s) { dst.handling = s.grid_boundary_handling; dst.penalty = s.penalty_stiffness; dst.flip_interior = s.flip_interior != 0; };
    if (kind == 0) for (uint32_t i = 0; i < n && i < o.cuboids.size(); ++i) put(o.cuboids[i].opts, opt[i]);
    if (kind == 1) for (uint32_t i = 0; i < n && i < o.heightfields.size(); ++i) put(o.heightfields[i].opts, opt[i]);
    if (kind == 2) for (uint32_t i = 0; i < n && i < o.trimeshes.size(); ++i) { put(o.trimeshes[i].opts, opt[i]); o.trimeshes[i].flip_interior = opt[i].flip_interior != 0; }
    if (kind == 3) for (uint32_t i = 0; i < n && i < o.polylines.size(); ++i) put(o.polylines[i].opts, opt[i]);
}
extern "C" {

void* oracle_create(int use_f64, float cell_width, uint32_t domain_log2_bytes) {
    Handle* h = new Handle;
    h->f64 = use_f64 != 0;
    if (h->f64) h->od = new Oracle<double>((double)cell_width, domain_log2_bytes);
    else h->of = new Oracle<float>(cell_width, domain_log2_bytes);
    if ((h->f64 && !h->od->mem) || (!h->f64 && !h->of->mem)) { delete h->od; delete h->of; delete h; return nullptr; }
    return h;
}
void oracle_destroy(void* hv) { Handle* h = (Handle*)hv; if (!h) return; delete h->of; delete h->od; delete h; }
void oracle_set_penalty_force(void* hv, int on) { Handle* h = (Handle*)hv; if (h->f64) h->od->penalty_force = on != 0; else h->of->penalty_force = on != 0; }
void oracle_set_num_threads(void* hv, int n) { Handle* h = (Handle*)hv; if (h->f64) h->od->num_threads = n; else h->of->num_threads = n; }

void oracle_set_particles(void* hv, uint64_t n, const SparklParticleStatus* st, const SparklParticlePosition* pos,
                          const SparklParticleVelocity* vel, const SparklParticleVolume* vol, const SparklParticlePhase* ph,
                          const SparklParticleData* data, const SparklVelocityGradient* vg) {
    Handle* h = (Handle*)hv;
    if (h->f64) set_particles(*h->od, n, st, pos, vel, vol, ph, data, vg); else set_particles(*h->of, n, st, pos, vel, vol, ph, data, vg);
}
uint64_t oracle_num_particles(void* hv) { Handle* h = (Handle*)hv; return h->f64 ? h->od->particles.size() : h->of->particles.size(); }
void oracle_get_particles(void* hv, SparklParticleStatus* st, SparklParticlePosition* pos, SparklParticleVelocity* vel,
                          SparklParticleVolume* vol, SparklParticlePhase* ph, SparklParticleData* data, SparklVelocityGradient* vg) {
    Handle* h = (Handle*)hv;
    if (h->f64) get_particles(*h->od, st, pos, vel, vol, ph, data, vg); else get_particles(*h->of, st, pos, vel, vol, ph, data, vg);
}
void oracle_get_state_f64(void* hv, double* pos, double* vel, double* F, double* C) {
    Handle* h = (Handle*)hv;
    if (h->f64) get_state_f64(*h->od, pos, vel, F, C); else get_state_f64(*h->of, pos, vel, F, C);
}
void oracle_set_models(void* hv, uint32_t n, const SparklParticleModel* m) {
    Handle* h = (Handle*)hv;
    if (h->f64) h->od->models.assign(m, m + n); else h->of->models.assign(m, m + n);
}
void oracle_set_cuboids(void* hv, uint32_t n, const SparklCuboidCollider* c) {
    Handle* h = (Handle*)hv;
    if (h->f64) set_cuboids(*h->od, n, c); else set_cuboids(*h->of, n, c);
}

void oracle_set_heightfields(void* hv, uint32_t n, const SparklHeightFieldCollider* c) {
    Handle* h = (Handle*)hv;
    if (h->f64) set_heightfields(*h->od, n, c); else set_heightfields(*h->of, n, c);
}
void oracle_set_polylines(void* hv, uint32_t n, const SparklPolylineCollider* c) {
    Handle* h = (Handle*)hv;
    if (h->f64) set_polylines(*h->od, n, c); else set_polylines(*h->of, n, c);
}
void oracle_set_trimeshes(void* hv, uint32_t n, const SparklTriMeshCollider* c) {
    Handle* h = (Handle*)hv;
    if (h->f64) set_trimeshes(*h->od, n, c); else set_trimeshes(*h->of, n, c);
}
void oracle_set_collider_options(void* hv, int kind, uint32_t n, const SparklColliderOptions* opt) {
    Handle* h = (Handle*)hv;
    if (h->f64) set_options(*h->od, kind, n, opt); else set_options(*h->of, kind, n, opt);
}
void oracle_trimesh_project(const SparklTriMeshCollider* c, const double pt[3], double out[3], int* is_inside) {
    TriMesh<double> m;
    fill_trimesh(m, *c);
    Projection<double> p = trimesh_project_point(m, Vec3<double>{{pt[0], pt[1], pt[2]}});
    for (int a = 0; a < 3; ++a) out[a] = p.point[a];
    *is_inside = p.is_inside;
}
void oracle_heightfield_project(const SparklHeightFieldCollider* c, const double pt[3], double out[3], int* is_inside) {
    Oracle<double>* dummy = nullptr; (void)dummy;
    HeightField<double> h;
    h.nrows = (int)c->nrows; h.ncols = (int)c->ncols;
    h.heights.assign(c->heights, c->heights + (size_t)c->nrows * c->ncols);
    for (int a = 0; a < 3; ++a) { h.scale[a] = c->scale[a]; h.t[a] = c->translation[a]; }
    for (int a = 0; a < 4; ++a) h.q[a] = c->rotation[a];
    h.friction = c->friction; h.below_is_solid = c->below_is_solid != 0;
    Projection<double> p = heightfield_project_point(h, Vec3<double>{{pt[0], pt[1], pt[2]}});
    for (int a = 0; a < 3; ++a) out[a] = p.point[a];
    *is_inside = p.is_inside;
}

// ParticleFracture (src_core/dynamics/particle.rs:180-195): crack_propagation_factor, crack_threshold per particle, caller order
void oracle_set_fracture(void* hv, uint64_t n, const float* cpf_and_threshold) {
    Handle* h = (Handle*)hv;
    for (uint64_t i = 0; i < n; ++i) {
        if (h->f64) { auto& p = h->od->particles[i]; p.crack_propagation_factor = cpf_and_threshold[2 * i]; p.crack_threshold = cpf_and_threshold[2 * i + 1]; }
        else { auto& p = h->of->particles[i]; p.crack_propagation_factor = cpf_and_threshold[2 * i]; p.crack_threshold = cpf_and_threshold[2 * i + 1]; }
    }
}
double oracle_step(void* hv, const SparklSolverParams* params, const float g[3]) {
    Handle* h = (Handle*)hv;
    if (h->f64) return h->od->step(*params, Vec3<double>{{g[0], g[1], g[2]}});
    return h->of->step(*params, Vec3<float>{{g[0], g[1], g[2]}});
}
int oracle_last_num_substeps(void* hv) { Handle* h = (Handle*)hv; return h->f64 ? h->od->last_num_substeps : h->of->last_num_substeps; }
void oracle_last_dts(void* hv, double* out) {
    Handle* h = (Handle*)hv;
    const auto& v = h->f64 ? h->od->last_dts : h->of->last_dts;
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
}
// one substep at a fixed dt (sort -> [fluid volumes] -> p2g -> grid update -> g2p)
double oracle_substep_fixed(void* hv, const SparklSolverParams* params, const float g[3], double dt) {
    Handle* h = (Handle*)hv;
    if (h->f64) return h->od->substep(*params, Vec3<double>{{g[0], g[1], g[2]}}, 0.0, 0.0, dt);
    return h->of->substep(*params, Vec3<float>{{g[0], g[1], g[2]}}, 0.0f, 0.0f, (float)dt);
}
// phase-by-phase access
void oracle_sort(void* hv) { Handle* h = (Handle*)hv; if (h->f64) h->od->sort(); else h->of->sort(); }
void oracle_recompute_fluids_volumes(void* hv) { Handle* h = (Handle*)hv; if (h->f64) h->od->recompute_fluids_volumes(); else h->of->recompute_fluids_volumes(); }
double oracle_estimate_dt(void* hv, double max_dt) {
    Handle* h = (Handle*)hv;
    return h->f64 ? h->od->adaptive_timestep_length(max_dt) : (double)h->of->adaptive_timestep_length((float)max_dt);
}
void oracle_p2g(void* hv, double dt, const float g[3]) {
    Handle* h = (Handle*)hv;
    if (h->f64) h->od->p2g(dt, Vec3<double>{{g[0], g[1], g[2]}}); else h->of->p2g((float)dt, Vec3<float>{{g[0], g[1], g[2]}});
}
void oracle_grid_update(void* hv, double dt, int boundary_handling, uint32_t dofs) {
    Handle* h = (Handle*)hv;
    if (h->f64) h->od->grid_update(dt, boundary_handling, dofs); else h->of->grid_update((float)dt, boundary_handling, dofs);
}
void oracle_g2p(void* hv, double dt) { Handle* h = (Handle*)hv; if (h->f64) h->od->g2p(dt); else h->of->g2p((float)dt); }
uint64_t oracle_get_nodes(void* hv, SparklGridNode* out, uint64_t cap) {
    Handle* h = (Handle*)hv;
    return h->f64 ? get_nodes(*h->od, out, cap) : get_nodes(*h->of, out, cap);
}
uint64_t oracle_num_regions(void* hv) { Handle* h = (Handle*)hv; return h->f64 ? h->od->regions.size() : h->of->regions.size(); }
uint64_t oracle_num_particle_regions(void* hv) {
    Handle* h = (Handle*)hv; uint64_t n = 0;
    if (h->f64) { for (auto& r : h->od->regions) n += r.end > r.begin; } else { for (auto& r : h->of->regions) n += r.end > r.begin; }
    return n;
}
uint64_t oracle_num_bins(void* hv) { Handle* h = (Handle*)hv; return h->f64 ? h->od->bins.size() : h->of->bins.size(); }
void oracle_get_order(void* hv, uint64_t* out) {
    Handle* h = (Handle*)hv;
    const auto& o = h->f64 ? h->od->order : h->of->order;
    for (size_t i = 0; i < o.size(); ++i) out[i] = o[i];
}
void oracle_get_grid_indices(void* hv, uint64_t* out) {
    Handle* h = (Handle*)hv;
    if (h->f64) { for (size_t i = 0; i < h->od->particles.size(); ++i) out[i] = h->od->particles[i].grid_index; }
    else { for (size_t i = 0; i < h->of->particles.size(); ++i) out[i] = h->of->particles[i].grid_index; }
}

// the stable parallel merge sort of ParticleSet::sort on caller-supplied keys (tests/test_oracle_kats.py checks it against numpy's stable argsort):
// out_idx[i] = index of the i-th key in stable order
void oracle_debug_parallel_sort(const uint64_t* keys, uint64_t n, int threads, uint64_t* out_idx) {
    std::vector<KeyIdx, no_init_allocator<KeyIdx>> a(n), tmp;
    for (uint64_t i = 0; i < n; ++i) a[i] = KeyIdx{keys[i], (size_t)i};
    parallel_stable_sort_pairs(a, tmp, threads);
    for (uint64_t i = 0; i < n; ++i) out_idx[i] = a[i].idx;
}

// ---- slab-decomposition checker hooks (tests/test_slab_gloo.py): region-wise access to (mass, momentum)
uint64_t oracle_get_region_ids(void* hv, uint64_t* out, uint64_t cap) {
    Handle* h = (Handle*)hv;
    uint64_t n = 0;
    auto put = [&](uint64_t id) { if (out && n < cap) out[n] = id; ++n; };
    if (h->f64) { for (auto& r : h->od->regions) put(r.id); } else { for (auto& r : h->of->regions) put(r.id); }
    return n;
}
// nodes[64][4] = (mass, px, py, pz), node i at region_id | (i << 6)
int oracle_get_block_nodes(void* hv, uint64_t region_id, float* nodes) {
    Handle* h = (Handle*)hv;
    for (uint64_t i = 0; i < 64; ++i) {
        uint64_t id = region_id | (i << PACK_ALIGN);
        if (h->f64) { auto& g = h->od->node(id); nodes[4 * i] = (float)g.mass; for (int a = 0; a < 3; ++a) nodes[4 * i + 1 + a] = (float)g.momentum[a]; }
        else { auto& g = h->of->node(id); nodes[4 * i] = g.mass; for (int a = 0; a < 3; ++a) nodes[4 * i + 1 + a] = g.momentum[a]; }
    }
    return 1;
}
int oracle_add_block_nodes(void* hv, uint64_t region_id, const float* nodes) {
    Handle* h = (Handle*)hv;
    if (h->f64 ? !region_active(*h->od, region_id) : !region_active(*h->of, region_id)) return 0;  // no local particle touches it
    for (uint64_t i = 0; i < 64; ++i) {
        uint64_t id = region_id | (i << PACK_ALIGN);
        if (h->f64) { auto& g = h->od->node(id); g.mass += nodes[4 * i]; for (int a = 0; a < 3; ++a) g.momentum[a] += nodes[4 * i + 1 + a]; }
        else { auto& g = h->of->node(id); g.mass += nodes[4 * i]; for (int a = 0; a < 3; ++a) g.momentum[a] += nodes[4 * i + 1 + a]; }
    }
    return 1;
}
// velocity = (momentum + mass g dt) inv_exact(mass) on every active node (particle_to_grid.rs:88-89)
void oracle_refresh_velocities(void* hv, double dt, const float g[3]) {
    Handle* h = (Handle*)hv;
    auto go = [&](auto& o) {
        using R = decltype(o.h);
        for (auto& r : o.regions)
            for (uint64_t i = 0; i < 64; ++i) {
                auto& c = o.node(r.id | (i << PACK_ALIGN));
                R im = inv_exact(c.mass);
                for (int a = 0; a < 3; ++a) c.velocity[a] = (c.momentum[a] + c.mass * (R)g[a] * (R)dt) * im;
            }
    };
    if (h->f64) go(*h->od); else go(*h->of);
}

// ---- SPGrid addressing (pinned by the reference's own tests, sp_grid.rs:411-555)
uint64_t oracle_pack(uint32_t x, uint32_t y, uint32_t z) { return pack(x, y, z); }
void oracle_unpack(uint64_t id, uint32_t out[3]) { unpack(id, out); }
uint64_t oracle_packed_add(uint64_t a, uint64_t b) { return packed_add(a, b); }
uint64_t oracle_morton_encode3(uint32_t x, uint32_t y, uint32_t z) { return morton_encode3(x, y, z); }
void oracle_nbh_shift(int k, int out[3]) { for (int a = 0; a < 3; ++a) out[a] = NBH_SHIFTS[k][a]; }
uint64_t oracle_packed_nbh_shift(int k) { return packed_nbh_shift(k); }
uint64_t oracle_packed_region_shift(int k) { return pack(NBH_REGION_SHIFTS[k][0], NBH_REGION_SHIFTS[k][1], NBH_REGION_SHIFTS[k][2]); }
int oracle_region_color(uint64_t region_id) { for (int c = 0; c < 8; ++c) if (Oracle<float>::region_has_color(region_id, c)) return c; return -1; }
// vectorised forms so the exhaustive reference tests run in C, not in Python loops
// returns the number of mismatches of unpack(pack(c)) == c over [x0,x1) x [y0,y1) x [z0,z1)
uint64_t oracle_check_pack_roundtrip(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1) {
    uint64_t bad = 0;
    for (uint32_t i = x0; i < x1; ++i) for (uint32_t j = y0; j < y1; ++j) for (uint32_t k = z0; k < z1; ++k) {
        uint32_t u[3]; unpack(pack(i, j, k), u);
        bad += !(u[0] == i && u[1] == j && u[2] == k);
    }
    return bad;
}
// test_morton_sum (sp_grid.rs:502-529): packed_add(pack(c), PACKED_NBH_SHIFTS[s]) == pack(c + NBH_SHIFTS[s])
uint64_t oracle_check_packed_add(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1) {
    uint64_t bad = 0;
    for (uint32_t i = x0; i < x1; ++i) for (uint32_t j = y0; j < y1; ++j) for (uint32_t k = z0; k < z1; ++k) {
        uint64_t packed = pack(i, j, k);
        for (int s = 0; s < 27; ++s) {
            uint32_t u[3]; unpack(packed_add(packed, packed_nbh_shift(s)), u);
            bad += !(u[0] == i + NBH_SHIFTS[s][0] && u[1] == j + NBH_SHIFTS[s][1] && u[2] == k + NBH_SHIFTS[s][2]);
        }
    }
    return bad;
}

// ---- unit-level model math (double in/out; computed in the requested precision)
void oracle_eval_all(int f64, double x, double out[3]) {
    if (f64) { double o[3]; eval_all<double>(x, o); for (int i = 0; i < 3; ++i) out[i] = o[i]; }
    else { float o[3]; eval_all<float>((float)x, o); for (int i = 0; i < 3; ++i) out[i] = o[i]; }
}
void oracle_svd3(const double A[9], double U[9], double S[3], double V[9]) { svd3_f64(A, U, S, V); }

void oracle_kirchhoff_stress(int f64, const SparklConstitutiveModel* cm, double phase, double hardening, const double F[9], double mass,
                             double volume0, double fluid_j, const double C[9], double out[9]) {
    if (f64) stress_impl<double>(cm, phase, hardening, F, mass, volume0, fluid_j, C, out);
    else stress_impl<float>(cm, phase, hardening, F, mass, volume0, fluid_j, C, out);
}
double oracle_pos_energy(int f64, const SparklConstitutiveModel* cm, double phase, double hardening, const double F[9]) {
    if (f64) {
        Mat3<double> f; for (int i = 0; i < 9; ++i) f.m[i] = F[i];
        if (cm->tag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR) return corotated_pos_energy<double>(cm->u.corotated, f, hardening);
        if (cm->tag == SPARKL_CONSTITUTIVE_NEO_HOOKEAN) return neo_hookean_pos_energy<double>(cm->u.neo_hookean, phase, hardening, f);
        return 0;
    }
    Mat3<float> f; for (int i = 0; i < 9; ++i) f.m[i] = (float)F[i];
    if (cm->tag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR) return corotated_pos_energy<float>(cm->u.corotated, f, (float)hardening);
    if (cm->tag == SPARKL_CONSTITUTIVE_NEO_HOOKEAN) return neo_hookean_pos_energy<float>(cm->u.neo_hookean, (float)phase, (float)hardening, f);
    return 0;
}
double oracle_dp_alpha(int f64, const SparklDruckerPragerPlasticity* m, double q) {
    return f64 ? dp_alpha<double>(*m, q) : (double)dp_alpha<float>(*m, (float)q);
}
// state = {Jp, q, log_vol_gain}
void oracle_dp_update(int f64, const SparklDruckerPragerPlasticity* m, double phase, double F[9], double state[3]) {
    if (f64) {
        Mat3<double> f; for (int i = 0; i < 9; ++i) f.m[i] = F[i];
        dp_update_particle<double>(*m, phase, f, state[0], state[1], state[2]);
        for (int i = 0; i < 9; ++i) F[i] = f.m[i];
    } else {
        Mat3<float> f; for (int i = 0; i < 9; ++i) f.m[i] = (float)F[i];
        float jp = (float)state[0], q = (float)state[1], lvg = (float)state[2];
        dp_update_particle<float>(*m, (float)phase, f, jp, q, lvg);
        for (int i = 0; i < 9; ++i) F[i] = f.m[i];
        state[0] = jp; state[1] = q; state[2] = lvg;
    }
}
void oracle_plastic_update(int f64, const SparklPlasticModel* pm, double phase, double F[9], double st[5]) {
    if (f64) plastic_update_t<double>(pm, phase, F, st); else plastic_update_t<float>(pm, phase, F, st);
}
int oracle_maximum_stress_failed(const SparklMaximumStressFailure* m, const double stress[9]) {
    Mat3<double> t; for (int i = 0; i < 9; ++i) t.m[i] = stress[i];
    return maximum_stress_failed<double>(*m, t) ? 1 : 0;
}
// returns 0 (None) / 1 (Some); out = {s0, s1, s2, dq}
int oracle_dp_project(int f64, const SparklDruckerPragerPlasticity* m, const double sv[3], double lvg, double alpha, double out[4]) {
    if (f64) {
        Vec3<double> o; double dq = 0;
        int r = dp_project<double>(*m, Vec3<double>{{sv[0], sv[1], sv[2]}}, lvg, alpha, o, dq);
        if (r) { out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = dq; }
        return r;
    }
    Vec3<float> o; float dq = 0;
    int r = dp_project<float>(*m, Vec3<float>{{(float)sv[0], (float)sv[1], (float)sv[2]}}, (float)lvg, (float)alpha, o, dq);
    if (r) { out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = dq; }
    return r;
}
void oracle_cuboid_project(const SparklCuboidCollider* c, const double pt[3], double out[3], int* is_inside) {
    Cuboid<double> cu;
    for (int a = 0; a < 3; ++a) { cu.he[a] = c->half_extents[a]; cu.t[a] = c->translation[a]; }
    for (int a = 0; a < 4; ++a) cu.q[a] = c->rotation[a];
    cu.friction = c->friction;
    Projection<double> p = cuboid_project_point(cu, Vec3<double>{{pt[0], pt[1], pt[2]}});
    for (int a = 0; a < 3; ++a) out[a] = p.point[a];
    *is_inside = p.is_inside;
}
double oracle_model_timestep_bound(void* hv, uint64_t particle) {
    Handle* h = (Handle*)hv;
    return h->f64 ? h->od->model_timestep_bound(h->od->particles[particle]) : (double)h->of->model_timestep_bound(h->of->particles[particle]);
}

// ---- timed run for bench.py's cpu_baseline leg: `nsub` fixed-dt substeps; returns seconds
double oracle_timed_substeps(void* hv, const SparklSolverParams* params, const float g[3], double dt, int nsub) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nsub; ++i) oracle_substep_fixed(hv, params, g, dt);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
