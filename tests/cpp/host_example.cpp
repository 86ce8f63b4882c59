// C++ host program against include/sparkl_hip.hpp: a sand3-like scene (examples3d/sand3.rs:39-78, without jitter) stepped
// through MpmPipeline::step and through fixed substeps.  Prints a small state digest that tests/test_cpp_host.py compares
// with the Python host mirror driving the same library.
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <tuple>

#include "sparkl_hip.hpp"

using namespace sparkl;

int main(int argc, char** argv) {
    const int nsub = argc > 1 ? atoi(argv[1]) : 10;
    try {
        const Real E = 1.0e7f, NU = 0.2f, cell_width = 0.2f, particle_rad = cell_width / 4.0f;
        ParticleModelSet models;
        auto sand_model = models.insert(ParticleModel::with_plasticity(CorotatedLinearElasticity(E, NU), DruckerPragerPlasticity(E, NU)));
        auto block_model = models.insert(ParticleModel::new_(CorotatedLinearElasticity(E, NU)));
        ParticleSet particles;
        const Real x0 = 4 * cell_width + particle_rad;
        particles.insert_batch(cube_particles({x0, x0 + 10 * particle_rad * 2.0f, x0}, 16, 10, 12, sand_model, particle_rad, 2700.0f));
        particles.insert_batch(cube_particles({x0, x0, x0}, 16, 10, 12, block_model, particle_rad, 2700.0f));
        SparklCuboidCollider floor{};
        floor.half_extents[0] = 1000; floor.half_extents[1] = 50; floor.half_extents[2] = 1000;
        floor.rotation[3] = 1.0f;
        floor.translation[1] = 4 * cell_width - 0.5f * cell_width - 50.0f;
        floor.friction = 0.5f;

        MpmPipeline pipeline(cell_width);
        pipeline.set_models(models);
        pipeline.set_colliders({floor});
        pipeline.upload(particles);
        SolverParameters params;
        const Vector gravity{0.0f, -9.81f, 0.0f};
        pipeline.substeps(params, gravity, 2.5e-4f, (uint32_t)nsub);
        SparklTimings tim{};
        const Real simulated = pipeline.step(params, gravity, &tim);
        pipeline.download(particles);
        double sx = 0, sy = 0, sz = 0, svy = 0, sf = 0;
        for (auto& p : particles.particles()) { sx += p.position[0]; sy += p.position[1]; sz += p.position[2]; svy += p.velocity[1]; sf += p.deformation_gradient[4]; }
        const double n = (double)particles.len();
        std::printf("n %zu simulated %.9g substeps %u mean_pos %.9g %.9g %.9g mean_vy %.9g mean_Fyy %.9g\n", particles.len(), simulated, tim.num_substeps, sx / n, sy / n, sz / n,
                    svy / n, sf / n);
        // heightfield + triangle-mesh colliders and the overlapped readback, through the same mirror
        {
            const float heights[9] = {0.f, 0.1f, 0.f, 0.1f, 0.2f, 0.1f, 0.f, 0.1f, 0.f};   // 3 x 3, column-major
            SparklHeightFieldCollider hf{};
            hf.nrows = 3; hf.ncols = 3; hf.heights = heights;
            hf.scale[0] = 20.f; hf.scale[1] = 1.f; hf.scale[2] = 20.f;
            hf.rotation[3] = 1.f; hf.translation[0] = 2.f; hf.translation[1] = 0.3f; hf.translation[2] = 2.f; hf.friction = 0.5f;
            const float verts[12] = {0.f, 0.f, 0.f, 4.f, 0.f, 0.f, 0.f, 0.f, 4.f, 0.f, 1.f, 0.f};
            const uint32_t idx[12] = {0, 2, 1, 0, 1, 3, 0, 3, 2, 1, 2, 3};
            SparklTriMeshCollider tm{};
            tm.num_vertices = 4; tm.vertices = verts; tm.num_triangles = 4; tm.indices = idx;
            tm.rotation[3] = 1.f; tm.translation[1] = -3.f; tm.friction = 0.5f; tm.oriented = 1;
            pipeline.set_colliders({});
            pipeline.set_heightfields({hf});
            pipeline.set_trimeshes({tm});
            const size_t np = particles.len();
            auto* pos = MpmPipeline::host_alloc<SparklParticlePosition>(np);
            auto* vel = MpmPipeline::host_alloc<SparklParticleVelocity>(np);
            pipeline.substeps(params, gravity, 2.5e-4f, 5);
            pipeline.readback_begin(pos, vel, nullptr);
            pipeline.substeps(params, gravity, 2.5e-4f, 5);    // overlaps the copies
            pipeline.readback_wait();
            double sy2 = 0;
            for (size_t i = 0; i < np; ++i) sy2 += pos[i].point[1];
            pipeline.download(particles);
            double sy3 = 0;
            for (auto& p : particles.particles()) sy3 += p.position[1];
            std::printf("readback: snapshot mean_y %.9g, five substeps later %.9g, %s\n", sy2 / (double)np, sy3 / (double)np, sy3 < sy2 ? "ok" : "NOT FALLING");
            MpmPipeline::host_free(pos); MpmPipeline::host_free(vel);
        }
        // two x-slabs of the same scene stepped together inside the library (sparkl_hip_group_substeps; between ranks the same loop
        // moves the ghost planes with RCCL): the union must equal the un-decomposed pipeline
        {
            ParticleSet all;
            all.insert_batch(cube_particles({x0, x0 + 10 * particle_rad * 2.0f, x0}, 16, 10, 12, sand_model, particle_rad, 2700.0f));
            all.insert_batch(cube_particles({x0, x0, x0}, 16, 10, 12, block_model, particle_rad, 2700.0f));
            const int32_t cut = 2;   // block index along x: cells 0..7 | 8..
            ParticleSet lo, hi;
            for (auto& p : all.particles()) {
                const int cell = (int)std::lround(p.position[0] / cell_width) - 1;
                ((cell >> 2) < cut ? lo : hi).insert(p);
            }
            MpmPipeline a(cell_width);
            MpmPipeline b(cell_width, 0, 30, a.stream());
            a.set_slab(0, INT32_MIN, cut, 0); b.set_slab(0, cut, INT32_MAX, lo.len());
            for (auto* m : {&a, &b}) { m->set_models(models); m->set_colliders({floor}); }
            a.upload(lo); b.upload(hi);
            MpmPipeline::group_substeps({&a, &b}, params, gravity, 2.5e-4f, 40);
            MpmPipeline whole(cell_width);
            whole.set_models(models); whole.set_colliders({floor}); whole.upload(all);
            whole.substeps(params, gravity, 2.5e-4f, 40);
            whole.download(all);
            double ref[2] = {0, 0}, got[2] = {0, 0};
            for (auto& p : all.particles()) { ref[0] += p.position[1]; ref[1] += p.velocity[1]; }
            size_t total = 0;
            for (auto* m : {&a, &b}) {
                std::vector<SparklParticlePosition> pos; std::vector<SparklParticleVelocity> vel; std::vector<uint32_t> ids;
                m->read_device_order(pos, vel, ids);
                total += pos.size();
                for (size_t i = 0; i < pos.size(); ++i) { got[0] += pos[i].point[1]; got[1] += vel[i].vector[1]; }
            }
            const bool ok = total == all.len() && std::fabs(got[0] - ref[0]) <= 1e-6 * std::fabs(ref[0]) && std::fabs(got[1] - ref[1]) <= 2e-3 * std::fabs(ref[1]);
            std::printf("slabs: %zu + %zu particles, sum_y %.9g vs %.9g, sum_vy %.9g vs %.9g, %s\n", lo.len(), hi.len(), got[0], ref[0], got[1], ref[1], ok ? "ok" : "MISMATCH");
        }
        // the reference's debug validator of the binning (check_active_blocks), through the library
        std::printf("grid check: %s\n", pipeline.grid_is_consistent() ? "ok" : "INCONSISTENT");
        // error behaviour: Custom models cannot cross the C ABI
        ParticleModelSet bad;
        ParticleModel custom;
        custom.core.constitutive_model.tag = SPARKL_CONSTITUTIVE_CUSTOM;
        bad.insert(custom);
        try { pipeline.set_models(bad); std::printf("custom: accepted (BUG)\n"); return 2; } catch (const Error& e) { std::printf("custom: rejected code %d\n", e.code); }
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
