//! Reference-fixture dumper for dimforge/sparkl (CPU dynamics path, `dim3` + `f32`).
//!
//! WRITTEN BLIND: the container this repository is built in has no rustc / cargo, so this file has never been compiled.  It only
//! uses sparkl's public API as read from the sources (file:line below) and `std`; expect to fix a typo or two.
//!
//! What it does: reads the scenes `tools/reference_fixtures/make_inputs.py` wrote (format: tools/reference_fixtures/format.md), steps
//! each one with sparkl's own `MpmPipeline::step` (src/pipelines/mpm_pipeline.rs:15-42 -> MlsSolver::step,
//! src/dynamics/solver/mls_solver.rs:15-139) at the fixed substep length of the scene, and writes the particle state after the listed
//! numbers of substeps as `<scene>_k<K>.state`.  Those files are REFERENCE-HELD VECTORS for the dynamics path -- the thing the oracle
//! of sparkl_amd (oracle/mpm_oracle.hpp) could not be pinned to so far: copy them to `tests/golden/reference/` of that repository and
//! `tests/test_reference_fixtures.py` holds the oracle (and on a GPU box the HIP path) to them.
//!
//! How to run it in a sparkl checkout:
//!   1. fix the mapping size typo, or every particle fails the domain test and G2P indexes out of bounds:
//!        src/geometry/sp_grid.rs:175   `MmapMut::map_anon(2 ^ 30)`  ->  `MmapMut::map_anon(1 << 30)`
//!   2. cp dump_fixtures.rs crates/sparkl3d/tests/dump_fixtures.rs
//!   3. SPARKL_FIXTURE_IN=<sparkl_amd>/tests/golden/reference/inputs SPARKL_FIXTURE_OUT=<sparkl_amd>/tests/golden/reference \
//!        cargo test -p sparkl3d --release --test dump_fixtures -- --nocapture
//!
//! One substep per `step()` call: `SolverParameters { dt, max_num_substeps: 1, .. }` makes min_dt = dt, and the scenes' dt sits below
//! every CFL bound (asserted: `step` must return exactly dt), so `adaptive_timestep_length` returns min(remaining, bounds) = dt
//! (mls_solver.rs:33-59).

use sparkl3d::na::{Matrix3, Point3, Vector3};
use sparkl3d::prelude::*;
use sparkl3d::rapier::geometry::{ColliderBuilder, ColliderSet};
use std::fs;
use std::io::Write;
use std::path::{Path, PathBuf};

struct SceneSpec {
    cell_width: f32,
    gravity: Vector3<f32>,
    dt: f32,
    substeps: Vec<u32>,
    boundary_handling: u32,
    force_fluids_volume_recomputation: bool,
    models: Vec<ParticleModel>,
    cuboids: Vec<[f32; 7]>,
}

fn parse_scene(path: &Path) -> SceneSpec {
    let text = fs::read_to_string(path).expect("scene file");
    let mut s = SceneSpec {
        cell_width: 0.0,
        gravity: Vector3::zeros(),
        dt: 0.0,
        substeps: vec![],
        boundary_handling: 1,
        force_fluids_volume_recomputation: false,
        models: vec![],
        cuboids: vec![],
    };
    for line in text.lines() {
        let line = line.trim();
        if line.is_empty() || line.starts_with('#') {
            continue;
        }
        let t: Vec<&str> = line.split_whitespace().collect();
        let f = |i: usize| -> f32 { t[i].parse::<f32>().expect("float field") };
        match t[0] {
            "cell_width" => s.cell_width = f(1),
            "gravity" => s.gravity = Vector3::new(f(1), f(2), f(3)),
            "dt" => s.dt = f(1),
            "substeps" => s.substeps = t[1..].iter().map(|x| x.parse().unwrap()).collect(),
            "boundary_handling" => s.boundary_handling = t[1].parse().unwrap(),
            "force_fluids_volume_recomputation" => s.force_fluids_volume_recomputation = t[1] != "0",
            "cuboid" => s.cuboids.push([f(1), f(2), f(3), f(4), f(5), f(6), f(7)]),
            "model" => {
                // every model struct has public fields (src_core/dynamics/models/*.rs): built from the raw numbers, no constructor arithmetic
                let (model, next): (ParticleModel, usize) = match t[1] {
                    "corotated" => (
                        ParticleModel::new(CorotatedLinearElasticity {
                            split_stress_on_failure: t[5] != "0",
                            cfl_coeff: f(4),
                            lambda: f(2),
                            mu: f(3),
                        }),
                        6,
                    ),
                    "neo_hookean" => (
                        ParticleModel::new(NeoHookeanElasticity {
                            cfl_coeff: f(4),
                            lambda: f(2),
                            mu: f(3),
                        }),
                        5,
                    ),
                    "monaghan" => (
                        ParticleModel::new(MonaghanSphEos {
                            pressure0: f(2),
                            gamma: t[3].parse().unwrap(),
                            viscosity: f(4),
                            max_neg_pressure: f(5),
                        }),
                        6,
                    ),
                    other => panic!("unknown constitutive model {}", other),
                };
                let mut model = model;
                if t.len() > next && t[next] == "dp" {
                    let b = next + 1;
                    model.plastic_model = Some(std::sync::Arc::new(DruckerPragerPlasticity {
                        h0: f(b),
                        h1: f(b + 1),
                        h2: f(b + 2),
                        h3: f(b + 3),
                        lambda: f(b + 4),
                        mu: f(b + 5),
                        only_active_when_failed: t[b + 6] != "0",
                        volume_correction: f(b + 7),
                    }));
                }
                s.models.push(model);
            }
            other => panic!("unknown scene record {}", other),
        }
    }
    s
}

fn read_f32s(raw: &[u8], off: &mut usize, count: usize) -> Vec<f32> {
    let out = (0..count)
        .map(|i| f32::from_le_bytes(raw[*off + 4 * i..*off + 4 * i + 4].try_into().unwrap()))
        .collect();
    *off += 4 * count;
    out
}

fn write_f32s(out: &mut Vec<u8>, vals: impl Iterator<Item = f32>) {
    for v in vals {
        out.extend_from_slice(&v.to_le_bytes());
    }
}

fn dump_state(path: &Path, particles: &ParticleSet, k: u32, dt: f32) {
    // caller (insertion) order: ParticleSet::particles() is never permuted, only `order` is (particle_set.rs:127-145)
    let ps = particles.particles();
    let mut out: Vec<u8> = Vec::new();
    out.extend_from_slice(b"SPKFIX01");
    out.extend_from_slice(&(ps.len() as u64).to_le_bytes());
    out.extend_from_slice(&k.to_le_bytes());
    out.extend_from_slice(&dt.to_le_bytes());
    write_f32s(&mut out, ps.iter().flat_map(|p| p.position.coords.iter().copied().collect::<Vec<_>>()));
    write_f32s(&mut out, ps.iter().flat_map(|p| p.velocity.iter().copied().collect::<Vec<_>>()));
    // Matrix3::as_slice() is nalgebra's storage order: column-major
    write_f32s(&mut out, ps.iter().flat_map(|p| p.deformation_gradient.as_slice().to_vec()));
    write_f32s(&mut out, ps.iter().flat_map(|p| p.velocity_gradient.as_slice().to_vec()));
    write_f32s(&mut out, ps.iter().map(|p| p.plastic_deformation_gradient_det));
    write_f32s(&mut out, ps.iter().map(|p| p.plastic_hardening));
    write_f32s(&mut out, ps.iter().map(|p| p.elastic_hardening));
    write_f32s(&mut out, ps.iter().map(|p| p.log_vol_gain));
    write_f32s(&mut out, ps.iter().map(|p| p.psi_pos));
    write_f32s(&mut out, ps.iter().map(|p| p.phase));
    out.extend(ps.iter().map(|p| p.failed as u8));
    fs::File::create(path).and_then(|mut f| f.write_all(&out)).expect("write state");
}

fn run_scene(name: &str, dir_in: &Path, dir_out: &Path) {
    let spec = parse_scene(&dir_in.join(format!("{}.scene", name)));
    let raw = fs::read(dir_in.join(format!("{}.input", name))).expect("input file");
    assert_eq!(&raw[..8], b"SPKINP01");
    let n = u64::from_le_bytes(raw[8..16].try_into().unwrap()) as usize;
    let mut off = 16;
    let model_idx: Vec<u32> = (0..n)
        .map(|i| u32::from_le_bytes(raw[off + 4 * i..off + 4 * i + 4].try_into().unwrap()))
        .collect();
    off += 4 * n;
    let pos = read_f32s(&raw, &mut off, 3 * n);
    let vel = read_f32s(&raw, &mut off, 3 * n);
    let mass = read_f32s(&raw, &mut off, n);
    let volume0 = read_f32s(&raw, &mut off, n);
    let radius0 = read_f32s(&raw, &mut off, n);
    assert_eq!(off, raw.len());

    let mut models = ParticleModelSet::new();
    let handles: Vec<ParticleModelHandle> = spec.models.iter().map(|m| models.insert(m.clone())).collect();

    let mut list = Vec::with_capacity(n);
    for i in 0..n {
        // Particle::new (src/dynamics/particle.rs:53-103) for every default, then the exact f32 inputs of the scene file
        let mut p = Particle::new(
            handles[model_idx[i] as usize],
            Point3::new(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
            radius0[i],
            mass[i] / volume0[i],
        );
        p.mass = mass[i];
        p.volume0 = volume0[i];
        p.radius0 = radius0[i];
        p.velocity = Vector3::new(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]);
        list.push(p);
    }
    let mut particles = ParticleSet::new();
    particles.insert_batch(list); // particle_set.rs:66-70

    let mut colliders = ColliderSet::new();
    for c in &spec.cuboids {
        colliders.insert(
            ColliderBuilder::cuboid(c[0], c[1], c[2])
                .translation(Vector3::new(c[3], c[4], c[5]))
                .friction(c[6])
                .build(),
        );
    }

    let mut grid = SpGrid::<GridNode>::new(spec.cell_width).expect("SpGrid (did you fix `2 ^ 30`?)");
    let mut grid_phase = SpGrid::<GridNodeCgPhase>::new(spec.cell_width).expect("SpGrid");
    let mut pipeline = MpmPipeline::new();
    let mut params = SolverParameters::default(); // src_core/dynamics/solver/solver_parameters.rs:55-68
    params.dt = spec.dt;
    params.max_num_substeps = 1;
    params.boundary_handling = match spec.boundary_handling {
        0 => BoundaryHandling::Stick,
        1 => BoundaryHandling::Friction,
        2 => BoundaryHandling::FrictionZUp,
        _ => BoundaryHandling::None,
    };
    params.force_fluids_volume_recomputation = spec.force_fluids_volume_recomputation;

    let last = *spec.substeps.iter().max().unwrap();
    for k in 1..=last {
        let rigid_world = RigidWorld { colliders: &colliders };
        let simulated = pipeline.step(
            &params,
            &spec.gravity,
            &rigid_world,
            &mut grid,
            &mut grid_phase,
            &mut particles,
            &models,
            &mut (),
        );
        assert_eq!(simulated, spec.dt, "substep {} of {}: the CFL bound cut the step short", k, name);
        if spec.substeps.contains(&k) {
            dump_state(&dir_out.join(format!("{}_k{}.state", name, k)), &particles, k, spec.dt);
        }
    }
    println!("{}: {} particles, {} substeps", name, n, last);
}

#[test]
fn dump_fixtures() {
    let dir_in = PathBuf::from(std::env::var("SPARKL_FIXTURE_IN").expect("SPARKL_FIXTURE_IN = directory of the .scene / .input files"));
    let dir_out = PathBuf::from(std::env::var("SPARKL_FIXTURE_OUT").expect("SPARKL_FIXTURE_OUT = where the .state files go"));
    fs::create_dir_all(&dir_out).unwrap();
    let mut names: Vec<String> = fs::read_dir(&dir_in)
        .unwrap()
        .filter_map(|e| {
            let p = e.unwrap().path();
            if p.extension().map_or(false, |x| x == "scene") {
                Some(p.file_stem().unwrap().to_string_lossy().into_owned())
            } else {
                None
            }
        })
        .collect();
    names.sort();
    assert!(!names.is_empty(), "no .scene files in {:?}", dir_in);
    for name in names {
        run_scene(&name, &dir_in, &dir_out);
    }
}
