// C++ host program against include/sparkl_hip.hpp: a sand3-like scene (examples3d/sand3.rs:39-78, without jitter) stepped
// through MpmPipeline::step and through fixed substeps.  Prints a small state digest that tests/test_cpp_host.py compares
// with the Python host mirror driving the same library.
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
