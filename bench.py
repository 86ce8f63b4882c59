#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the MI355X MPM hot path on BASELINE.json's configurations.

One "step" = one substep (P2G -> grid update -> G2P, re-bin when a particle leaves its tile margin) of fixed length on
synthetic, HBM-resident particles.  Prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config sand10m|neo1m|fluid20m|cube80m|snow1m] [--scaling weak|strong] [--particles P] [--no-cpu-baseline]

--config sand10m (default) is BASELINE.json's metric: the 10 M-particle Drucker-Prager column (configs[2]);
neo1m = configs[1] (1 M neo-Hookean cube), fluid20m = configs[3] (20 M fluid particles, volume recompute sweep),
cube80m = configs[4] (cube_through_sand3, ONE 80 M-particle scene: on N GPUs it is cut into N x-slabs -- strong scaling);
snow1m is no BASELINE configuration: a 1 M-particle snow block, the one scene that times the RARE kernel instantiations;
sandblock10m is configs[0]'s scene (sand3.rs: Drucker-Prager sand ON an elastic block) at 10 M particles: TWO materials, so the one-material
specialisations of the sand column (compile-time model view, 4-byte flags mirror) stand down -- the generic fused kernel, measured.
--scaling strong (default since round 6): the configured size is the WHOLE job whatever N -- BASELINE.json's metric is the 10 M column on the whole
node with ">= 6x strong scaling at 8 GPUs" -- sand10m at N > 1: 192 x 260 x 200 = 9.98 M cut into N x-slabs (N = 1: the 200 x 250 x 200 column as ever);
the weak-scaling figure (a 10 M column PER RANK) rides along as `weak_scaling` in the same line (--no-weak-figure: skip it).
--scaling weak: every rank owns a column of the configured size, the job grows with N (`value` is then that figure).
N = 1 also carries `strong_projection`: the 8-slab group loop of the same job on the ONE device, per slab -- a projection, not a scaling measurement.
--gpus N > 1 with no WORLD_SIZE in the environment starts N ranks by itself (one process per GPU, RCCL); under
torch.distributed.run the ranks come from the environment.  See DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# algorithmic bytes: per particle and per active node, by kernel
BYTES = {
    "sand": {"p2g": 112, "g2p_r": 64, "g2p_w": 108},
    "elastic": {"p2g": 112, "g2p_r": 52, "g2p_w": 96},
    "fluid": {"p2g": 76, "g2p_r": 20, "g2p_w": 64},
}
NODE_BYTES = {"p2g": 16, "grid_update": 28, "g2p": 12}
# config 4's volume-recompute sweep (+48 B/particle, +12 B/node): mass-only P2G 20 B, its gather 24 + 4 B
VOLUME_BYTES = {"p2g_mass": 20, "g2p_volume": 28}
VOLUME_NODE_BYTES = {"p2g_mass": 4, "grid_update_mass": 8}
# the parity gate of tests/parity.py (TOLERANCE_STATEMENT), stated beside every number this file prints
PARITY_TOLERANCE = ("arbiter = the reference's algorithm evaluated in f64 (oracle/): eps_x (in h) and eps_v (relative to max |v|) of the HIP path "
                    "vs that trajectory <= 1e-4 at every size (no storage allowance since round 5: positions are held relative to the home block); vs the "
                    "reference's own f32 arithmetic (f32 oracle): eps <= eps(f32 oracle vs f64 oracle) + 1e-4 for positions and velocities -- the f32 oracle "
                    "itself sits 1e-4..6e-4 from the f64 trajectory on stiff sand, 3.9e-4 in eps_x on the 10.8 m cube and 4.6e-3 in eps_v on the fluid dam")
KERNEL_NAMES = {"p2g": "k_p2g_pair", "g2p": "k_g2p", "grid_update": "k_grid_update", "g2p2g": "k_g2p2g", "p2g_mass": "k_p2g_mass",
                "grid_update_mass": "k_grid_update<mass>", "g2p_volume": "k_g2p_volume"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clock-wake-n", type=int, default=4096)
    ap.add_argument("--clock-wake-kind", choices=("matmul", "self"), default="matmul", help="what keeps the GPU busy during the wake-up: fp32 matmuls, or substeps of the same scene on a second pipeline (freed after the measurement)")
    ap.add_argument("--clock-wake-ms", type=float, default=250.0, help="untimed fp32 matmuls before the warmup substeps so that the run does not start in the power controller's load-onset clock dip (0 = off)")
    ap.add_argument("--no-cold-window", dest="cold_window", action="store_false",
                    help="skip the cold window: the same W + K substeps timed from an idle GPU BEFORE the clock wake-up (reported beside the headline, never as `value`)")
    ap.add_argument("--no-copy-peak", action="store_true", help="skip the 1 GiB device-to-device copy that measures the stream peak")
    ap.add_argument("--config", choices=("sand10m", "neo1m", "fluid20m", "cube80m", "snow1m", "sandblock10m"), default="sand10m")
    ap.add_argument("--particles", type=float, default=None, help="particles per GPU with --scaling weak, in the whole job with --scaling strong (default: the configuration's own size)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak (default for sand10m): every rank owns a column of the configured size, the job grows with N; strong (default and only mode of cube80m): "
                         "the whole job is the configured size -- 80 M for cube80m, 10 M for sand10m -- cut into N x-slabs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak-figure", dest="weak_figure", action="store_false", help="N > 1, strong scaling: do not also time the weak-scaling job (a 10 M column per rank)")
    ap.add_argument("--no-strong-projection", dest="strong_projection", action="store_false", help="N = 1, sand10m: skip the 8-slab group loop on the one device (`strong_projection`)")
    ap.add_argument("--no-kernel-clock", action="store_true", help="skip the per-kernel hipEvent spans")
    ap.add_argument("--no-late-phase", dest="late_phase", action="store_false", help="skip the flowing-phase measurements (late_collapse, deep_collapse of sand10m; late_collapse = the moving dam of fluid20m; N = 1)")
    args = ap.parse_args(argv)
    if args.scaling is None:
        # the job the metric names is fixed: the series N = 1, 2, 4, 8 the driver measures is a STRONG-scaling series (round 6; rounds 1 - 5 grew the job with N)
        args.scaling = "strong" if args.config in ("cube80m", "sand10m") else "weak"
    if args.config == "cube80m" and args.scaling != "strong":
        ap.error("cube80m is BASELINE configs[4]: ONE 80 M-particle scene cut into slabs (--scaling strong)")
    if args.scaling == "strong" and args.config not in ("cube80m", "sand10m"):
        ap.error("--scaling strong cuts the sand scenes (sand10m, cube80m) into x-slabs")
    return args


# ------------------------------------------------------------------------------------------ N > 1: launcher
def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) and relay rank 0's JSON line.

    The parent never touches the GPU (no torch import, no HIP call): it only spawns `python bench.py <same flags>` N times
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as torch.distributed.run would."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True, start_new_session=True))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(codes) or line is None:
        print(f"bench.py: rank exit codes {codes}" + ("" if line else "; rank 0 printed no result line"), file=sys.stderr)
        return 1
    print(line)
    return 0


# ------------------------------------------------------------------------------------------ workloads
def column_dims(n_particles):
    """200 x 250 x 200 at 10M (SURVEY config 3), scaled isotropically otherwise."""
    f = (n_particles / 10.0e6) ** (1.0 / 3.0)
    return max(8, int(round(200 * f))), max(8, int(round(250 * f))), max(8, int(round(200 * f)))


def strong_dims(total, world, base=(200, 250, 200)):
    """Lattice of ~`total` particles with the aspect of `base`, its x size a whole number of 4-cell blocks (8 particles) per rank
    for every world size up to 8 -- the SAME lattice whatever N, so that the job is fixed."""
    f = (total / float(base[0] * base[1] * base[2])) ** (1.0 / 3.0)
    nx = max(64, int(round(base[0] * f / 64.0)) * 64)
    nz = max(8, int(round(base[2] * f)))
    ny = max(8, int(round(total / float(nx * nz))))
    assert nx % (8 * world) == 0, f"{world} ranks do not divide the {nx // 8} block planes of the scene"
    return nx, ny, nz


def build_scene(args, rank, world, mode):
    """Returns (scene, material key of BYTES, description, x-lattice size of this rank's slab)."""
    from sparkl_amd import scenes
    slab = (rank, world) if mode == "slabs" else (0, 1)
    if args.config == "cube80m":
        # BASELINE configs[4]: cube_through_sand3 (examples3d/cube_through_sand3.rs:40-81), 800 x 250 x 400 bed = 80 M + the kinematic cube
        total = args.particles or 80.0e6
        nx, ny, nz = strong_dims(total, world, base=(800, 250, 400))
        cube_n = max(2, int(round(25 * (total / 80.0e6) ** (1.0 / 3.0))))
        sc = scenes.cube_through_sand(nx // slab[1], ny, nz, cube_n=cube_n, name="config5_cube_through_sand", x_slab=slab)
        desc = (f"cube_through_sand3: Drucker-Prager sand bed {nx}x{ny}x{nz} = {nx * ny * nz} particles + kinematic cube {cube_n}^3 at (10,0,0) m/s "
                f"(8 ppc, h=0.2, E=1e7 nu=0.2 rho=2700, cuboid friction floor), fixed dt={sc.fixed_dt}; {len(sc.particles)} particles on this rank")
        return sc, "sand", desc, nx // slab[1]
    if args.config == "sand10m" and args.scaling == "strong" and world > 1:
        nx, ny, nz = strong_dims(args.particles or 10.0e6, world)
        sc = scenes.sand_column(nx // slab[1], ny, nz, name="config3_drucker_prager_column", x_slab=slab)
        desc = (f"3D Drucker-Prager sand column collapse, {nx}x{ny}x{nz} = {nx * ny * nz} particles in the whole job, {len(sc.particles)} on this rank (8 ppc, h=0.2, "
                f"E=1e7 nu=0.2 rho=2700, cuboid friction floor), fixed dt={sc.fixed_dt}")
        return sc, "sand", desc, nx // slab[1]
    if args.config == "sand10m":
        nx, ny, nz = column_dims(args.particles or 10.0e6)
        if mode == "slabs":
            nx = max(8, int(round(nx / 8.0)) * 8)          # whole 4-cell blocks per slab (200 at the 10M default)
        sc = scenes.sand_column(nx, ny, nz, name="config3_drucker_prager_column", x_slab=slab)
        desc = (f"3D Drucker-Prager sand column collapse, {len(sc.particles)} particles/GPU ({nx}x{ny}x{nz} lattice, 8 ppc, h=0.2, "
                f"E=1e7 nu=0.2 rho=2700, cuboid friction floor), fixed dt={sc.fixed_dt}")
        return sc, "sand", desc, nx
    if mode == "slabs":
        raise SystemExit("--gpus > 1 runs the slab decomposition of the sand scenes only (sand10m = BASELINE configs[2], cube80m = configs[4])")
    if args.config == "sandblock10m":
        # configs[0]'s scene at size (examples3d/sand3.rs:39-78: a Drucker-Prager sand layer on a corotated elastic block of the same footprint): two models in
        # the table, so k_g2p2g runs its GENERIC instantiation (model dispatch per particle, the 16-byte k record) -- what the column's one-material
        # specialisations (MSET = 1 / 4) are worth is the difference to the sand10m line
        f = ((args.particles or 10.0e6) / 10.0e6) ** (1.0 / 3.0)
        nx, ny, nz = max(8, int(round(200 * f))), max(4, int(round(125 * f))), max(8, int(round(200 * f)))
        sc = scenes.sand_column(nx, ny, nz, with_block=True, name="config1_sand3_at_10m")
        desc = (f"sand3.rs at size: Drucker-Prager sand {nx}x{ny}x{nz} on a corotated elastic block {nx}x{ny}x{nz} = {len(sc.particles)} particles (8 ppc, h=0.2, E=1e7 nu=0.2 "
                f"rho=2700 both, cuboid friction floor), fixed dt={sc.fixed_dt}: two models -> the generic fused kernel; not BASELINE.json's headline configuration")
        return sc, "sand", desc, nx
    if args.config == "snow1m":
        # not a BASELINE config: the RARE instantiations of the particle kernels (snow / Rankine / NACC plasticity, failure model, particle
        # projection, penalty force) measured on one scene of their own -- a 1 M-particle snow block dropped on the floor
        # (plasticity_snow.rs; the scene of tests/test_gpu_parity.py::test_rare_material_models_parity at size)
        from sparkl_amd.models import CorotatedLinearElasticity, SnowPlasticity
        n = max(8, int(round((args.particles or 1.0e6) ** (1.0 / 3.0))))
        sc = scenes.block_on_floor(n, CorotatedLinearElasticity(1.4e5, 0.2), SnowPlasticity(), density=400.0, v0=(0.0, -8.0, 0.0), dt=2.0e-4, name="snow_block")
        desc = (f"snow block (corotated E=1.4e5 nu=0.2 + SnowPlasticity, rho=400) {n}^3 = {len(sc.particles)} particles dropped on a friction floor at 8 m/s "
                f"(8 ppc, h=0.2), fixed dt={sc.fixed_dt}: the RARE kernel instantiations; not a BASELINE.json configuration")
        return sc, "sand", desc, n
    if args.config == "neo1m":
        n = max(8, int(round((args.particles or 1.0e6) ** (1.0 / 3.0))))
        sc = scenes.elastic_cube(n, name="config2_neo_hookean_cube")
        desc = (f"3D neo-Hookean elastic cube, {len(sc.particles)} particles ({n}^3 lattice, 8 ppc, h=0.2, E=1e6 nu=0.3 rho=1000, no gravity / collider, "
                f"spinning + expanding), fixed dt={sc.fixed_dt}")
        return sc, "elastic", desc, n
    f = ((args.particles or 20.1e6) / 20.1e6) ** (1.0 / 3.0)
    nx, ny, nz = max(8, int(round(344 * f))), max(8, int(round(240 * f))), max(8, int(round(244 * f)))
    sc = scenes.fluid_dam(nx, ny, nz, name="config4_fluid_dam")
    desc = (f"3D weakly-compressible fluid dam break (Monaghan EOS p0=1e6 gamma=7, rho=1000), {len(sc.particles)} particles ({nx}x{ny}x{nz} lattice, r=0.1, h=0.8: "
            f"64 ppc), force_fluids_volume_recomputation, frictionless cuboid floor, fixed dt={sc.fixed_dt}")
    return sc, "fluid", desc, nx


def cpu_baseline(max_threads, sc=None, desc=None):
    """The oracle (restated sparkl CPU path, f32, OpenMP, `kind: "port"`) on the benchmark's OWN configuration: 2 substeps of
    the scene the GPU just ran (the 10 M column at the default), timed phase by phase (sort / P2G / grid update / G2P as
    BASELINE.md section 2 lists them) after one untimed substep (first touch of the mmap'd grid).

    Like the reference (serial region loop, particle_set.rs:171-253) the port stops scaling early: the thread count is probed
    on one substep of a 2.5 M-particle column each (round 6; a 400 k column until round 5, small enough to sit in the last-level caches of one
    socket and to pick 16 threads for a 10 M scene that wants more), every probe on an oracle of its own, so that particles AND grid pages are
    first touched by the thread team that then works on them (the grid in the untimed substep); the best count is used and reported as `cores`."""
    from sparkl_amd import scenes
    from oracle import Oracle
    nx, ny, nz = column_dims(2_500_000)
    probe = scenes.sand_column(nx, ny, nz, name="cpu_thread_probe")
    n = len(probe.particles)
    best_t, best_rate, rates = 1, 0.0, {}
    for t in (8, 16, 32, 64, 96, 128, 192, 256):
        if t > max_threads:
            break
        o = Oracle(probe.cell_width, domain_log2_bytes=probe.domain_log2_bytes, threads=t)
        o.load_scene(probe)
        o.substep_fixed(probe.params, probe.gravity, probe.fixed_dt)   # first touch of the mmap'd grid
        rate = n / o.timed_substeps(probe.params, probe.gravity, probe.fixed_dt, 1)
        o.close()
        rates[t] = rate
        if rate > best_rate:
            best_t, best_rate = t, rate
    if sc is None:
        sc, desc = probe, f"{n}-particle Drucker-Prager column ({nx}x{ny}x{nz})"
    n = len(sc.particles)
    o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, threads=best_t)
    o.load_scene(sc)
    dt = sc.fixed_dt
    o.substep_fixed(sc.params, sc.gravity, dt)
    nsub = 2 if n > 2_000_000 else int(max(2, min(50, 10.0 * best_rate / n)))
    phases = {"sort": 0.0, "p2g": 0.0, "grid_update": 0.0, "g2p": 0.0}
    if sc.params.force_fluids_volume_recomputation:
        phases["fluid_volumes"] = 0.0
    t_all = time.perf_counter()
    for _ in range(nsub):   # the order of oracle_substep_fixed (mls_solver.rs:33-139 at a fixed step length)
        t0 = time.perf_counter(); o.sort(); phases["sort"] += time.perf_counter() - t0
        if "fluid_volumes" in phases:
            t0 = time.perf_counter(); o.recompute_fluids_volumes(); phases["fluid_volumes"] += time.perf_counter() - t0
        t0 = time.perf_counter(); o.p2g(dt, sc.gravity); phases["p2g"] += time.perf_counter() - t0
        t0 = time.perf_counter(); o.grid_update(dt, int(sc.params.boundary_handling), int(sc.params.simulation_dofs)); phases["grid_update"] += time.perf_counter() - t0
        t0 = time.perf_counter(); o.g2p(dt); phases["g2p"] += time.perf_counter() - t0
    t = time.perf_counter() - t_all
    o.close()
    return {"value": n * nsub / t, "unit": "particle-steps/s", "cores": best_t, "kind": "port",
            "ms_per_substep": 1e3 * t / nsub, "phase_ms_per_substep": {k: 1e3 * v / nsub for k, v in phases.items()},
            "thread_probe_particle_steps_per_s": {str(k): v for k, v in rates.items()},
            "sample": f"{nsub} substeps (after one untimed) of the benchmark's own configuration: {desc}; f32 oracle (oracle/mpm_oracle.hpp: restated sparkl CPU "
                      f"path), OpenMP with {best_t} threads = the best of 8/16/32/64/96/128/192/256 probed on a {len(probe.particles)}-particle column (one substep each, first touch by the probing team), on a {max_threads}-thread host"}


def recorded_traffic(kernel, n_particles, config):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (they cannot run inside the timed bench);
    only reported for the configuration and particle count they were recorded at."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    rec = rec.get(config, rec if config == "sand10m" else {})
    if abs(rec.get("particles_per_gpu", 0) - n_particles) > 0.001 * n_particles or kernel not in rec.get("bytes_per_launch", {}):
        return None, None
    return rec["bytes_per_launch"][kernel], rec["source"] + "; " + rec["correction"]


def kernel_bytes(material, n_particles, nodes, fluid_volumes):
    b = BYTES[material]
    kb = {
        "p2g": n_particles * b["p2g"] + nodes * NODE_BYTES["p2g"],
        "grid_update": nodes * NODE_BYTES["grid_update"],
        "g2p": n_particles * (b["g2p_r"] + b["g2p_w"]) + nodes * NODE_BYTES["g2p"],
    }
    # fixed dt: G2P of substep s and P2G of substep s+1 run as ONE kernel (k_g2p2g); its algorithmic bytes are the sum
    kb["g2p2g"] = kb["p2g"] + kb["g2p"]
    step = kb["p2g"] + kb["grid_update"] + kb["g2p"]
    if fluid_volumes:
        kb["p2g_mass"] = n_particles * VOLUME_BYTES["p2g_mass"] + nodes * VOLUME_NODE_BYTES["p2g_mass"]
        kb["grid_update_mass"] = nodes * VOLUME_NODE_BYTES["grid_update_mass"]
        kb["g2p_volume"] = n_particles * VOLUME_BYTES["g2p_volume"]
        step += kb["p2g_mass"] + kb["grid_update_mass"] + kb["g2p_volume"]
        # fixed-dt batches run the sweep fused: k_vol_p2g = volume gather + P2G, k_g2p_mass = G2P + the next mass scatter
        kb["p2g"] += kb["g2p_volume"]
        kb["g2p"] += kb["p2g_mass"]
    return kb, step


def clock_spans(clock):
    spans = {"p2g": (clock.p2g_ms, clock.p2g_launches), "grid_update": (clock.grid_update_ms, clock.grid_update_launches),
             "g2p": (clock.g2p_ms, clock.g2p_launches), "g2p2g": (clock.g2p2g_ms, clock.g2p2g_launches)}
    for name in ("p2g_mass", "grid_update_mass", "g2p_volume"):
        if hasattr(clock, name + "_ms"):
            spans[name] = (getattr(clock, name + "_ms"), getattr(clock, name + "_launches"))
    return {k: v for k, v in spans.items() if v[1] > 0}


def roofline_of(clock, kb, n_particles, config, kernel=None):
    """`roofline` object for the kernel the span set spends most of its time in (or the named one)."""
    spans = clock_spans(clock)
    if not spans:
        return None, {}
    avg_ms = {name: ms / n for name, (ms, n) in spans.items()}
    dom = kernel if kernel in spans else max(spans, key=lambda k: spans[k][0])
    achieved = kb[dom] / (avg_ms[dom] * 1e-3) / 1e9
    names = dict(KERNEL_NAMES)
    if "g2p_volume" in kb:
        names.update(p2g="k_vol_p2g", g2p="k_g2p_mass")
    traffic, traffic_src = recorded_traffic(names[dom].split("<")[0] if dom != "grid_update_mass" else "k_grid_update_mass", n_particles, config)
    return {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": kb[dom], "avg_launch_ms": avg_ms[dom],
            "launches_sampled": int(spans[dom][1])}, avg_ms


def scenes_mod():
    from sparkl_amd import scenes
    return scenes


def strong_projection(args, stream, device_id, single_ms, slabs=8, steps=96, warmup=32):
    """The strong-scaling job (192 x 260 x 200 = 9.98 M, bench.py --gpus 8) as 8 slab pipelines of THIS device stepped by sparkl_hip_group_substeps."""
    import torch
    from sparkl_amd.pipeline import MpmPipeline
    from sparkl_amd.slab import INT32_MAX, INT32_MIN
    nx, ny, nz = strong_dims(10.0e6, slabs)
    pipes, total = [], 0
    for r in range(slabs):
        s = scenes_mod().sand_column(nx // slabs, ny, nz, x_slab=(r, slabs))
        p = MpmPipeline(s.cell_width, device_id=device_id, domain_log2_bytes=s.domain_log2_bytes, stream=stream.cuda_stream)
        blocks = (nx // slabs) // 8
        p.set_slab(0, INT32_MIN if r == 0 else 1 + r * blocks, INT32_MAX if r == slabs - 1 else 1 + (r + 1) * blocks, id_offset=r * len(s.particles))
        p.load_scene(s)
        pipes.append(p)
        total += len(s.particles)
        params, gravity, dt = s.params, s.gravity, s.fixed_dt
    try:
        MpmPipeline.group_substeps(pipes, params, gravity, dt, warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        MpmPipeline.group_substeps(pipes, params, gravity, dt, steps)
        torch.cuda.synchronize()
        group_ms = 1e3 * (time.perf_counter() - t0) / steps
        info = pipes[0].direct_info()
        rebins = [int(p.group_rebins()) for p in pipes]
    finally:
        for p in pipes:
            p.close()
    per_slab_ms = group_ms / slabs
    return {"what": f"PROJECTION, not a measurement of scaling: the strong-scaling job ({nx}x{ny}x{nz} = {total} particles) as {slabs} x-slabs on ONE device, "
                    f"{steps} substeps through sparkl_hip_group_substeps (one stream: the slabs run one after the other), against this run's single pipeline",
            "slabs": slabs, "particles_per_slab": total // slabs, "group_ms_per_substep_one_device": group_ms, "per_slab_ms_per_substep": per_slab_ms,
            "single_pipeline_ms_per_substep": single_ms, "fixed_cost_us_per_slab_substep": 1e3 * (group_ms - single_ms) / slabs,
            "projected_speedup_at_8_gpus": single_ms / per_slab_ms, "direct_exchange": info, "collective_rebins": rebins,
            "not_included": "the xGMI hop of the halo stores and of the flag mail (a few us per substep), clock effects of 8 GPUs running at a 12 % duty cycle each, load imbalance as the column collapses"}


# ------------------------------------------------------------------------------------------ one rank
def main(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    result_fd = None
    if world > 1:
        # a rank that waits forever in a collective (a peer died, a link is down) must not hold the node: dump the stacks
        # and exit after 15 minutes (a whole 8-rank run takes well under two)
        import faulthandler
        faulthandler.dump_traceback_later(900, exit=True)
        # stdout carries ONE JSON line (rank 0's): gloo ("[Gloo] Rank 0 is connected to 7 peer ranks ...") and RCCL (its version banner) write
        # there from C++ whenever they please, so everything but that line goes to stderr for the whole run
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
    # test hook: SPARKL_BENCH_BACKEND=gloo + SPARKL_BENCH_ONE_DEVICE=1 runs the N-rank slab path on a single GPU
    backend = os.environ.get("SPARKL_BENCH_BACKEND", "nccl")
    device_id = 0 if os.environ.get("SPARKL_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(device_id)
    ranks = 1
    # N > 1 transports: "rccl" (default) = the exchange inside libsparkl_hip.so (sparkl_hip_group_substeps: RCCL send / recv of the
    # ghost planes, flag all-reduce, migration); torch.distributed (gloo, CPU) only bootstraps the 128-byte communicator id and
    # takes the MAX of the elapsed times.  "torch" = the Python SlabStepper over torch.distributed (nccl, or gloo for the
    # one-GPU test hook, where two ranks cannot open one device twice in RCCL).
    comm_mode = os.environ.get("SPARKL_BENCH_COMM", "rccl" if backend == "nccl" else "torch")
    if world > 1:
        import torch.distributed as dist
        default_pg_is_cpu = comm_mode == "rccl" or backend != "nccl"   # gloo carries the bootstrap / timing reductions then
        if comm_mode == "rccl":
            dist.init_process_group("gloo")
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_id))
        else:
            dist.init_process_group(backend)
        if comm_mode != "rccl":
            # the rank count the job really runs with: an all-reduce of ones over the transport that carries the halo
            ones = torch.ones(1, device="cuda" if backend == "nccl" else "cpu", dtype=torch.int64)
            dist.all_reduce(ones)
            ranks = int(ones.item())
            assert ranks == dist.get_world_size() == world, (ranks, dist.get_world_size(), world)

    from sparkl_amd.pipeline import MpmPipeline
    from sparkl_amd.slab import INT32_MAX, INT32_MIN, HipSlabBackend, SlabStepper

    # weak scaling: rank k owns the k-th x-slab of a column `world` times longer (config 5's decomposition);
    # each slab holds `particles` particles, cut on block planes, the shared node planes swapped every substep
    mode = os.environ.get("SPARKL_BENCH_MODE", "slabs" if world > 1 else "single")
    t0 = time.time()
    sc, material, desc, nx = build_scene(args, rank, world, mode)
    n_particles = len(sc.particles)
    gen_s = time.time() - t0
    dt = sc.fixed_dt
    fluid_volumes = bool(sc.params.force_fluids_volume_recomputation) and material == "fluid"
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    pipe = MpmPipeline(sc.cell_width, device_id=device_id, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
    stepper = None
    if mode == "slabs":
        cells_per_rank = nx // 2                       # lattice spacing h/2
        assert cells_per_rank % 4 == 0, "slab width must be a whole number of 4-cell blocks"
        first_block = 1                                # the column starts at cell 4
        lo = INT32_MIN if rank == 0 else first_block + rank * (cells_per_rank // 4)
        hi = INT32_MAX if rank == world - 1 else first_block + (rank + 1) * (cells_per_rank // 4)
        pipe.set_slab(0, lo, hi, id_offset=getattr(sc, "id_offset", rank * n_particles))
    t0 = time.time()
    pipe.load_scene(sc)
    upload_s = time.time() - t0
    if os.environ.get("SPARKL_BENCH_REORDER_HEADS"):   # experiments: in-place reorder threshold in run heads per particle (library default 0.6)
        pipe.reorder_policy_set(float(os.environ["SPARKL_BENCH_REORDER_HEADS"]) * n_particles, 0.0)
    in_library = mode == "slabs" and comm_mode == "rccl"
    group = None
    if in_library:
        ok = 1
        try:
            ids = [MpmPipeline.comm_unique_id() if rank == 0 else None]
        except Exception as e:                      # librccl missing or unusable: every rank sees the same library
            print(f"bench.py: in-library RCCL unavailable ({e}); falling back to torch.distributed", file=sys.stderr)
            ids, ok = [None], 0
        dist.broadcast_object_list(ids, src=0)
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if ok and ids[0] is not None:
                pipe.comm_init(ids[0], rank, world)
                ranks = pipe.comm_allreduce_sum(1)      # counted over the RCCL communicator that carries the halo
            else:
                ok = 0
        except Exception as e:
            print(f"bench.py: rank {rank}: sparkl_hip_comm_init failed ({e}); falling back to torch.distributed", file=sys.stderr)
            ok = 0
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        # all ranks use the in-library path or none does
        agree = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 1:
            assert ranks == dist.get_world_size() == world, (ranks, dist.get_world_size(), world)
        else:
            in_library = False
            comm_mode = "torch"
            group = dist.new_group(backend="nccl") if backend == "nccl" else None
            ones = torch.ones(1, device="cuda" if backend == "nccl" else "cpu", dtype=torch.int64)
            dist.all_reduce(ones, group=group)
            ranks = int(ones.item())
    if mode == "slabs" and not in_library:
        stepper = SlabStepper(HipSlabBackend(pipe, torch.device("cuda", device_id)), rank, world, group=group, chunk=16)

    def run(n):
        if in_library:
            MpmPipeline.group_substeps([pipe], sc.params, sc.gravity, dt, n)
        elif stepper is not None:
            stepper.substeps(sc.params, sc.gravity, dt, n)
        else:
            pipe.substeps(sc.params, sc.gravity, dt, n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pipe.sync()

    # Cold window (N = 1): what the same command reads WITHOUT the clock wake-up -- W untimed + K timed substeps from an idle GPU, i.e. inside
    # the power controller's load-onset dip -- measured first, reported as `cold_window` beside the headline.  The scene is then uploaded
    # again, so the headline window below covers the same substeps of the same initial state.
    cold = None
    wake_pipe = None
    if world == 1 and args.cold_window and args.clock_wake_ms > 0:
        run(args.warmup)
        cstride = 4 if args.steps >= 16 else (2 if args.steps >= 4 else 1)
        if not args.no_kernel_clock:
            pipe.clock_enable(True, stride=cstride)
            pipe.clock_read(reset=True)
        barrier()
        tc = time.perf_counter()
        run(args.steps)
        barrier()
        cold_s = time.perf_counter() - tc
        cclock = None if args.no_kernel_clock else pipe.clock_read(reset=True)
        pipe.clock_enable(False)
        ckb, cstep_bytes = kernel_bytes(material, n_particles, pipe.num_active_blocks() * 64, fluid_volumes)
        cold = {"steps": args.steps, "warmup": args.warmup, "clock_wake_ms": 0.0, "ms_per_step": 1e3 * cold_s / args.steps,
                "value": n_particles * args.steps / cold_s, "step_frac_of_hbm_peak": cstep_bytes / (cold_s / args.steps) / 1e9 / HBM_PEAK_GBS}
        if cclock is not None and cclock.sampled_substeps > 0:
            cold["roofline"], cold["kernels_avg_launch_ms"] = roofline_of(cclock, ckb, n_particles, args.config)
        pipe.load_scene(sc)
        pipe.sync()
    if args.clock_wake_ms > 0:
        # DVFS wake-up, untimed and before the W warmup substeps.  From idle, the first heavy launches make the power
        # controller drop the shader clock from 2.4 to ~1.7 GHz and walk it back up over ~35 ms (sampled with a one-wave
        # clock64 / wall_clock64 kernel: tools/debug/clock_probe.py, profiles/r02c_clock_probe.txt) -- exactly where the
        # driver's `--warmup 5 --steps 20` window (3..15 ms after the first launch) sits.  A GPU that is already busy does not
        # dip: a few plain fp32 matmuls on the bench stream are enough.  Nothing of the workload is skipped or moved.
        try:
            if args.clock_wake_kind == "self":
                wake_pipe = MpmPipeline(sc.cell_width, device_id=device_id, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
                wake_pipe.load_scene(sc)
                t_end = time.perf_counter() + args.clock_wake_ms * 1e-3
                while time.perf_counter() < t_end:
                    wake_pipe.substeps(sc.params, sc.gravity, dt, 32)
                raise StopIteration
            with torch.cuda.stream(stream):
                a = torch.randn(args.clock_wake_n, args.clock_wake_n, device="cuda")
                b = torch.randn(args.clock_wake_n, args.clock_wake_n, device="cuda")
                t_end = time.perf_counter() + args.clock_wake_ms * 1e-3
                while time.perf_counter() < t_end:
                    for _ in range(4):
                        c = a @ b
                    stream.synchronize()
                del a, b, c
        except StopIteration:
            pass
        except Exception as e:   # the wake-up is a convenience: a missing BLAS must not cost the measurement
            print(f"bench.py: clock wake-up skipped ({e})", file=sys.stderr)
            args.clock_wake_ms = 0.0
    if world > 1 and in_library:
        # The exchange the group loop keeps is timed on the job's own hardware (peer-direct stores, staged in line, staged overlapped: one undisturbed batch each,
        # the fastest stays -- sparkl_hip.hip, direct_cal_state): let that happen HERE, untimed, and start the measured substeps from the initial state again.
        run(112)
        pipe.load_scene(sc)
        pipe.sync()
    run(args.warmup)
    # event spans on every 4th substep of the timed region (every 2nd of a region shorter than 16 substeps): a recorded event is a
    # barrier packet of its own between two kernels that would otherwise be dispatched back to back -- sampling every launch of the
    # driver's 20-substep window cost it 0.008 - 0.012 ms per substep (profiles/r03_kernel_experiments.txt section 12; attaching the events to the dispatch
    # with hipExtLaunchKernelGGL costs more, 0.018: tried)
    stride = 4 if args.steps >= 16 else (2 if args.steps >= 4 else 1)
    if not args.no_kernel_clock:
        pipe.clock_enable(True, stride=stride)
        pipe.clock_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)   # EXACTLY K substeps; synchronises at the end
    barrier()
    elapsed = time.perf_counter() - t0
    if wake_pipe is not None:   # (--clock-wake-kind self: freeing it before the window would have been an idle gap)
        wake_pipe.close()
        wake_pipe = None
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if default_pg_is_cpu else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_particles = n_particles
    if world > 1:   # ranks of a strong-scaling job own different counts (the kinematic cube sits on rank 0; slabs re-balance)
        tn = torch.tensor([n_particles], device="cpu" if default_pg_is_cpu else "cuda", dtype=torch.int64)
        dist.all_reduce(tn)
        total_particles = int(tn.item())
    clock = None if args.no_kernel_clock else pipe.clock_read(reset=True)
    n_blocks, n_pblocks = pipe.num_active_blocks(), pipe.num_particle_blocks()
    kb, step_bytes = kernel_bytes(material, n_particles, n_blocks * 64, fluid_volumes)

    # The same column once it flows (cells hold uneven particle counts, re-bins and in-place reorders happen): outside the
    # timed region and never part of `value`, but measured the same way -- wall clock over 128 substeps and the dominant
    # kernel's hipEvent spans -> `roofline_late` (see profiles/r01_reorder_policy.txt)
    late, deep = None, None
    if world == 1 and args.late_phase and args.config in ("sand10m", "fluid20m"):
        substeps_run = [args.steps + args.warmup]

        def phase(start):
            """128 substeps from substep `start` on, timed like the headline region (wall clock + the dominant kernel's spans)."""
            before = max(0, start - substeps_run[0])
            pipe.clock_enable(False)
            run(before)
            if not args.no_kernel_clock:
                pipe.clock_enable(True, stride=4)
                pipe.clock_read(reset=True)
            pipe.sync()
            t1 = time.perf_counter()
            run(128)
            pipe.sync()
            ms = 1e3 * (time.perf_counter() - t1) / 128
            out = {"substeps_before": substeps_run[0] + before, "substeps_timed": 128, "ms_per_step": ms}
            substeps_run[0] += before + 128
            if not args.no_kernel_clock:
                lclock = pipe.clock_read(reset=True)
                nb = pipe.num_active_blocks()
                kb_l, step_bytes_l = kernel_bytes(material, n_particles, nb * 64, fluid_volumes)
                out["roofline"], avg = roofline_of(lclock, kb_l, n_particles, args.config, kernel="g2p2g" if args.config == "sand10m" else None)
                out["kernels_avg_launch_ms"] = avg
                out["step_frac_of_hbm_peak"] = step_bytes_l / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                out["resorts"] = int(lclock.resorts)
                out["noop_launches"] = int(lclock.noop_launches)   # launches behind a re-bin request (returned at once): not in the averages
                out["reorder_ms_total"] = lclock.reorder_ms
                out["active_node_blocks"] = nb
            return out

        if args.config == "sand10m":
            late = phase(640)      # 0.16 s after release: the column flows
            deep = phase(2000)     # 0.5 s: deep in the collapse (cells hold uneven counts, the foot spreads, re-bins every ~100 substeps)
        else:
            late = phase(2000)     # the 20 M dam 0.2 s after release (dt = 1e-4): the front runs, cells hold odd counts, re-bins and reorders happen

    # N > 1, strong scaling of the 10 M column (the default): the weak-scaling job -- a 10 M column PER RANK, what rounds 1 - 5 reported as `value` --
    # timed the same way on a second set of pipelines, reported as `weak_scaling` beside the headline
    weak = None
    if world > 1 and mode == "slabs" and args.config == "sand10m" and args.scaling == "strong" and args.weak_figure:
        wnx, wny, wnz = column_dims(args.particles or 10.0e6)
        wnx = max(8, int(round(wnx / 8.0)) * 8)
        wsc = scenes_mod().sand_column(wnx, wny, wnz, name="config3_drucker_prager_column_weak", x_slab=(rank, world))
        wpipe = MpmPipeline(wsc.cell_width, device_id=device_id, domain_log2_bytes=wsc.domain_log2_bytes, stream=stream.cuda_stream)
        wpipe.set_slab(0, INT32_MIN if rank == 0 else 1 + rank * (wnx // 8), INT32_MAX if rank == world - 1 else 1 + (rank + 1) * (wnx // 8), id_offset=rank * len(wsc.particles))
        wpipe.load_scene(wsc)
        wstep = None
        if in_library:
            wid = [MpmPipeline.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(wid, src=0)
            sys.stdout.flush(); saved = os.dup(1); os.dup2(2, 1)
            try:
                wpipe.comm_init(wid[0], rank, world)
            finally:
                sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
        else:
            wstep = SlabStepper(HipSlabBackend(wpipe, torch.device("cuda", device_id)), rank, world, group=group, chunk=16)

        def wrun(n):
            if in_library:
                MpmPipeline.group_substeps([wpipe], wsc.params, wsc.gravity, wsc.fixed_dt, n)
            else:
                wstep.substeps(wsc.params, wsc.gravity, wsc.fixed_dt, n)
        if in_library:   # (the exchange calibration of this set of pipelines: see above)
            wrun(112)
            wpipe.load_scene(wsc)
            wpipe.sync()
        wrun(args.warmup)
        dist.barrier(); torch.cuda.synchronize(); wpipe.sync()
        tw = time.perf_counter()
        wrun(args.steps)
        dist.barrier(); torch.cuda.synchronize(); wpipe.sync()
        welapsed = time.perf_counter() - tw
        tt = torch.tensor([welapsed], device="cpu" if default_pg_is_cpu else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wtotal = torch.tensor([len(wsc.particles)], device="cpu" if default_pg_is_cpu else "cuda", dtype=torch.int64)
        dist.all_reduce(wtotal)
        weak = {"scaling": "weak", "particles_total": int(wtotal.item()), "particles_per_gpu": len(wsc.particles), "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * float(tt.item()) / args.steps, "value": int(wtotal.item()) * args.steps / float(tt.item()), "unit": "particle-steps/s",
                "direct_exchange": wpipe.direct_info(),
                "note": "one 10 M column per rank (the job grows with N): what `value` was in rounds 1 - 5; never the headline"}
        wpipe.close()

    # N = 1: what the SAME 10 M job costs per slab when it is cut 8 ways -- the 8-member group loop of the library on this ONE device (peer-direct
    # exchange: k_halo_push, flag mail, grid update), one stream.  A PROJECTION of 8-GPU strong scaling from two measured numbers, not a measurement of it:
    # no xGMI hop, no second GPU has run.
    projection = None
    if world == 1 and args.config == "sand10m" and args.strong_projection and not args.particles:
        try:
            projection = strong_projection(args, stream, device_id, 1e3 * elapsed / args.steps)
        except Exception as e:   # a convenience figure: never at the price of the headline
            print(f"bench.py: strong_projection skipped ({e})", file=sys.stderr)

    # the achieved rate is also reported against a MEASURED stream peak, not only the 8 TB/s of the
    # data sheet: a device-to-device copy of 1 GiB (read + write = 2 GiB of traffic per copy), hipEvent-timed, after the run
    copy_gbs = None
    if rank == 0 and not args.no_copy_peak:
        with torch.cuda.stream(stream):
            src = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                dst.copy_(src)
            e1.record(stream)
            e1.synchronize()
            copy_gbs = 20 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst

    if rank == 0:
        value = total_particles * args.steps / elapsed
        headline = args.config == "sand10m" and (args.scaling == "strong" or world == 1)
        metric = "particle-steps/sec (whole node) + achieved HBM GB/s, 3D sand 10M particles" if headline else \
            f"particle-steps/sec (whole node) + achieved HBM GB/s, {args.config} {args.scaling} scaling (not BASELINE.json's headline metric)"
        out = {
            "metric": metric,
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "ranks": ranks,
            "config": {"workload": desc, "name": args.config, "particles_per_gpu": n_particles, "particles_total": total_particles,
                       "parity_tolerance": PARITY_TOLERANCE,
                       "active_node_blocks": n_blocks, "particle_blocks": n_pblocks,
                       "parallelism": "single GPU" if world == 1 else (
                           f"{world} x-slabs of one {world * nx}-particle-long scene ({'fixed total, strong scaling' if args.scaling == 'strong' else 'one configured column per rank, weak scaling'}), the four shared node planes of every cut swapped with both neighbours every substep "
                           f"({('peer-direct inside libsparkl_hip.so: k_halo_push stores into the hipIpc-mapped arenas of the neighbours, the re-bin flag travels as mail to every rank and is agreed in the grid update -- no per-substep collective' if (pipe.direct_info()['ready'] and pipe.direct_info()['chosen'] != 'staged') else 'RCCL send/recv inside libsparkl_hip.so (sparkl_hip_group_substeps), re-bin flag all-reduce') + '; the exchange was timed on this job before the warmup (direct_exchange.timed_us_per_substep), the fastest of peer-direct / staged in line / staged overlapped runs' if in_library else backend + ' send/recv from the Python SlabStepper, re-bin flag all-reduce'})"
                           if mode == "slabs" else f"{world} independent replicas")},
            "step_gbs": step_bytes * (total_particles / n_particles) / (elapsed / args.steps) / 1e9,
            "step_frac_of_hbm_peak": step_bytes * (total_particles / n_particles) / world / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
            "setup_s": {"generate": gen_s, "upload_and_pack": upload_s},
            "clock_wake_ms": args.clock_wake_ms, "clock_wake_kind": args.clock_wake_kind if args.clock_wake_ms > 0 else None,
        }
        if clock is not None and clock.sampled_substeps > 0:
            out["roofline"], avg_ms = roofline_of(clock, kb, n_particles, args.config)
            if copy_gbs:
                out["roofline"]["measured_copy_gbs"] = copy_gbs
                out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / copy_gbs
                if out["roofline"]["traffic"]:
                    # the bytes the kernel really moves (PMC) per second, against what a plain copy moves per second
                    out["roofline"]["traffic_gbs"] = out["roofline"]["traffic"] / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
                    out["roofline"]["traffic_frac_of_measured_copy"] = out["roofline"]["traffic_gbs"] / copy_gbs
            out["kernels_avg_launch_ms"] = avg_ms
            out["kernels_gbs"] = {name: kb[name] / (avg_ms[name] * 1e-3) / 1e9 for name in avg_ms}
            out["kernel_clock_sampled_substeps"] = int(clock.sampled_substeps)
            out["resort"] = {"count": int(clock.resorts), "ms_total": clock.sort_ms}
            if mode == "slabs":
                ns = max(int(clock.sampled_substeps), 1)
                out["halo_ms_per_step"] = clock.halo_ms / ns
                # what a substep of the slab loop costs beyond its particle / node kernels: wall clock per substep minus their spans per
                # sampled substep (halo pack / exchange / store, flag all-reduce, event hops, the host's look at the control block per
                # batch).  With the overlapped schedule the kernels of the two streams run side by side, so this is a lower bound there.
                # per kind: mean over its WORKING launches (no-op launches behind a re-bin request are counted apart, `noop_*`) times its launches
                # per sampled substep that did work -- dividing the sums by every sampled substep under-counts the kernels when no-ops occur
                grid_n = max(int(clock.grid_update_launches), 1)      # one working grid update per working substep
                kernel_ms = (clock.p2g_ms + clock.grid_update_ms + clock.g2p_ms + clock.g2p2g_ms + clock.p2g_mass_ms + clock.grid_update_mass_ms + clock.g2p_volume_ms) / min(ns, grid_n)
                out["slab_overhead_us"] = 1e3 * (out["ms_per_step"] - kernel_ms)
                chosen, us_inline, us_overlap = pipe.overlap_info()
                out["overlap"] = {"chosen": chosen, "in_line_us_per_substep": us_inline, "overlapped_us_per_substep": us_overlap,
                                  "note": "schedule of the halo chain between RCCL ranks, chosen on the MAX over the ranks of one 16-substep batch per schedule (-1: not decided / forced by SPARKL_HIP_OVERLAP)"}
        if mode == "slabs":
            out["direct_exchange"] = pipe.direct_info()
        if weak is not None:
            out["weak_scaling"] = weak
        if projection is not None:
            out["strong_projection"] = projection
        if cold is not None:
            out["cold_window"] = cold
        if late is not None:
            out["late_collapse"] = late
            if late.get("roofline"):
                out["roofline_late"] = late["roofline"]
        if deep is not None:
            out["deep_collapse"] = deep
            if deep.get("roofline"):
                out["roofline_deep"] = deep["roofline"]
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only: the other ranks would sit in the final barrier
            threads = len(os.sched_getaffinity(0))
            # the port does 10 M particles in ~4 s per substep; an 80 M scene would take minutes: the thread-probe column stands in
            out["cpu_baseline"] = cpu_baseline(threads, sc, desc) if n_particles <= 12_000_000 else cpu_baseline(threads)
        if result_fd is not None:
            sys.stdout.flush()
            os.write(result_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    _args = parse_args()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(_args))
    main(_args)
