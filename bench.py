#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the MI355X MPM hot path on BASELINE.json's 10M-particle Drucker-Prager column.

One "step" = one substep (P2G -> grid update -> G2P, re-bin when a particle leaves its tile margin) of fixed
length dt = 2.5e-4 on synthetic, HBM-resident particles.  Prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P] [--no-cpu-baseline]
N > 1 is launched by torch.distributed.run (one rank per GPU); see DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# algorithmic bytes: per particle and per active node, by kernel
BYTES = {
    "sand": {"p2g": 112, "g2p_r": 64, "g2p_w": 108},
    "elastic": {"p2g": 112, "g2p_r": 52, "g2p_w": 96},
    "fluid": {"p2g": 76, "g2p_r": 20, "g2p_w": 64},
}
NODE_BYTES = {"p2g": 16, "grid_update": 28, "g2p": 12}


def column_dims(n_particles):
    """200 x 250 x 200 at 10M (SURVEY config 3), scaled isotropically otherwise."""
    f = (n_particles / 10.0e6) ** (1.0 / 3.0)
    return max(8, int(round(200 * f))), max(8, int(round(250 * f))), max(8, int(round(200 * f)))


def cpu_baseline(max_threads, seconds_budget=20.0):
    """The oracle (restated sparkl CPU path, f32, OpenMP) on a bounded sample of the same workload.

    Like the reference (serial region loop, particle_set.rs:171-253) it stops scaling early: the thread count is probed
    on one substep each and the best one is used and reported."""
    from sparkl_amd import scenes
    from oracle import Oracle
    nx, ny, nz = column_dims(400_000)
    sc = scenes.sand_column(nx, ny, nz, name="cpu_sample")
    n = len(sc.particles)
    best_t, best_rate = 1, 0.0
    for t in (8, 16, 32, 64, 128):
        if t > max_threads:
            break
        o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, threads=t)
        o.load_scene(sc)
        o.substep_fixed(sc.params, sc.gravity, sc.fixed_dt)   # first touch of the mmap'd grid
        rate = n / o.timed_substeps(sc.params, sc.gravity, sc.fixed_dt, 1)
        o.close()
        if rate > best_rate:
            best_t, best_rate = t, rate
    o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, threads=best_t)
    o.load_scene(sc)
    o.substep_fixed(sc.params, sc.gravity, sc.fixed_dt)
    nsub = int(max(2, min(100, (seconds_budget - 5.0) * best_rate / n)))
    t = o.timed_substeps(sc.params, sc.gravity, sc.fixed_dt, nsub)
    o.close()
    return {"value": n * nsub / t, "unit": "particle-steps/s", "cores": best_t, "kind": "port",
            "sample": f"{n}-particle Drucker-Prager column ({nx}x{ny}x{nz}), {nsub} substeps of dt=2.5e-4, f32 oracle (oracle/mpm_oracle.hpp), OpenMP, "
                      f"best of 8/16/32/64/128 threads on a {max_threads}-thread host"}


def recorded_traffic(kernel, n_particles):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (they cannot run inside the timed bench);
    only reported for the particle count they were recorded at."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if abs(rec.get("particles_per_gpu", 0) - n_particles) > 0.001 * n_particles or kernel not in rec.get("bytes_per_launch", {}):
        return None, None
    return rec["bytes_per_launch"][kernel], rec["source"] + "; " + rec["correction"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=float, default=10.0e6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-clock", action="store_true", help="skip the per-kernel hipEvent spans")
    ap.add_argument("--no-late-phase", dest="late_phase", action="store_false", help="skip the informational flowing-phase measurement")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hook: SPARKL_BENCH_BACKEND=gloo + SPARKL_BENCH_ONE_DEVICE=1 runs the N-rank slab path on a single GPU
    backend = os.environ.get("SPARKL_BENCH_BACKEND", "nccl")
    device_id = 0 if os.environ.get("SPARKL_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(device_id)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_id))
        else:
            dist.init_process_group(backend)

    from sparkl_amd import scenes
    from sparkl_amd.pipeline import MpmPipeline
    from sparkl_amd.slab import INT32_MAX, INT32_MIN, HipSlabBackend, SlabStepper

    # weak scaling: rank k owns the k-th x-slab of a column `world` times longer (config 5's decomposition);
    # each slab holds `particles` particles, cut on block planes, ghost node-block planes swapped every substep
    nx, ny, nz = column_dims(args.particles)
    mode = os.environ.get("SPARKL_BENCH_MODE", "slabs" if world > 1 else "single")
    if mode == "slabs":
        nx = max(8, int(round(nx / 8.0)) * 8)          # whole 4-cell blocks per slab (200 at the 10M default)
    t0 = time.time()
    sc = scenes.sand_column(nx, ny, nz, name="config3_drucker_prager_column", x_slab=(rank, world) if mode == "slabs" else (0, 1))
    n_particles = len(sc.particles)
    gen_s = time.time() - t0
    dt = sc.fixed_dt
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
        pipe.set_slab(0, lo, hi, id_offset=rank * n_particles)
    t0 = time.time()
    pipe.load_scene(sc)
    upload_s = time.time() - t0
    if mode == "slabs":
        stepper = SlabStepper(HipSlabBackend(pipe, torch.device("cuda", device_id)), rank, world, chunk=16)

    def run(n):
        if stepper is not None:
            stepper.substeps(sc.params, sc.gravity, dt, n)
        else:
            pipe.substeps(sc.params, sc.gravity, dt, n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pipe.sync()

    run(args.warmup)
    if not args.no_kernel_clock:
        pipe.clock_enable(True, stride=4)   # event spans on every 4th substep: <1 % perturbation of the timed region
        pipe.clock_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)   # EXACTLY K substeps; synchronises at the end
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    clock = None if args.no_kernel_clock else pipe.clock_read(reset=False)
    n_blocks, n_pblocks = pipe.num_active_blocks(), pipe.num_particle_blocks()
    # informational, outside the timed region and never part of `value`: the same column once it flows (cells hold uneven
    # particle counts, re-bins and in-place reorders happen) -- see profiles/r01_reorder_policy.txt
    late = None
    if world == 1 and args.late_phase:
        run(max(0, 640 - args.steps - args.warmup))
        pipe.sync()
        t1 = time.perf_counter()
        run(128)
        pipe.sync()
        late = {"substeps_before": max(640, args.steps + args.warmup), "substeps_timed": 128, "ms_per_step": 1e3 * (time.perf_counter() - t1) / 128}

    if rank == 0:
        total_particles = n_particles * world
        value = total_particles * args.steps / elapsed
        nodes = n_blocks * 64
        b = BYTES["sand"]
        per_kernel_bytes = {
            "p2g": n_particles * b["p2g"] + nodes * NODE_BYTES["p2g"],
            "grid_update": nodes * NODE_BYTES["grid_update"],
            "g2p": n_particles * (b["g2p_r"] + b["g2p_w"]) + nodes * NODE_BYTES["g2p"],
        }
        step_bytes = sum(per_kernel_bytes.values())
        out = {
            "metric": "particle-steps/sec (whole node) + achieved HBM GB/s, 3D sand 10M particles",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D Drucker-Prager sand column collapse, {n_particles} particles/GPU ({nx}x{ny}x{nz} lattice, 8 ppc, h=0.2, "
                                   f"E=1e7 nu=0.2 rho=2700, cuboid friction floor), fixed dt={dt}", "particles_per_gpu": n_particles,
                       "active_node_blocks": n_blocks, "particle_blocks": n_pblocks,
                       "parallelism": "single GPU" if world == 1 else (
                           f"{world} x-slabs of one {world * nx}x{ny}x{nz} column, ghost node-block planes swapped with both neighbours every substep "
                           f"({backend} send/recv), re-bin flag all-reduce" if mode == "slabs" else f"{world} independent replicas")},
            "step_gbs": step_bytes * world / (elapsed / args.steps) / 1e9,
            "step_frac_of_hbm_peak": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
            "setup_s": {"generate": gen_s, "upload_and_pack": upload_s},
        }
        if clock is not None and clock.sampled_substeps > 0:
            # fixed dt: G2P of substep s and P2G of substep s+1 run as ONE kernel (k_g2p2g); its algorithmic bytes are the sum
            per_kernel_bytes["g2p2g"] = per_kernel_bytes["p2g"] + per_kernel_bytes["g2p"]
            spans = {"p2g": (clock.p2g_ms, clock.p2g_launches), "grid_update": (clock.grid_update_ms, clock.grid_update_launches),
                     "g2p": (clock.g2p_ms, clock.g2p_launches), "g2p2g": (clock.g2p2g_ms, clock.g2p2g_launches)}
            avg_ms = {name: ms / n for name, (ms, n) in spans.items() if n > 0}
            total_ms = {name: ms for name, (ms, n) in spans.items() if n > 0}
            dom = max(total_ms, key=total_ms.get)          # the kernel the timed region spends most of its time in
            achieved = per_kernel_bytes[dom] / (avg_ms[dom] * 1e-3) / 1e9
            kname = {"p2g": "k_p2g", "g2p": "k_g2p", "grid_update": "k_grid_update", "g2p2g": "k_g2p2g"}[dom]
            traffic, traffic_src = recorded_traffic(kname, n_particles)
            out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": per_kernel_bytes[dom],
                               "avg_launch_ms": avg_ms[dom], "launches_sampled": int(spans[dom][1])}
            out["kernels_avg_launch_ms"] = avg_ms
            out["kernels_gbs"] = {name: per_kernel_bytes[name] / (avg_ms[name] * 1e-3) / 1e9 for name in avg_ms}
            out["kernel_clock_sampled_substeps"] = int(clock.sampled_substeps)
            out["resort"] = {"count": int(clock.resorts), "ms_total": clock.sort_ms}
            if mode == "slabs":
                out["halo_ms_per_step"] = clock.halo_ms / max(int(clock.sampled_substeps), 1)
        if late is not None:
            out["late_collapse"] = late
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only: the other ranks would sit in the final barrier
            threads = len(os.sched_getaffinity(0))
            out["cpu_baseline"] = cpu_baseline(threads)
        print(json.dumps(out))
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
