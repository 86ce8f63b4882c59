"""One parameterised probe for the GPU box (python tools/probe.py <command> ...); replaces the one-off scripts of rounds 1 - 3
(tools/debug/*: late_probe, run_stats, soak*, slab_overhead, soak_slabs, step_vs_fixed, fixed_vs_adaptive, cold_start ...; their
findings are in profiles/HISTORY.md and profiles/r0N_*.txt).  Library variants: SPARKL_HIP_LIB=path/to/ab_x.so (tools/ab_build.sh).

  phase  [--config sand10m|neo1m|fluid20m|cube] [--marks 100 640 2000] [--steps 128] [--stats] [--reorder-heads H]
         run the scene to each mark, then time --steps substeps there: wall clock per substep, kernel spans, re-bins, block reorders;
         --stats prints the run structure the fused scatter sees (sparkl_hip_debug_run_stats, probes.h) before and after
  soak   [--config ...] [--chunks 8] [--chunk 500] [--step FRAME_DT]
         long run in chunks: throughput, kernel spans, re-bins, reorders, C rebuilds per chunk; validator + sanity check at the end;
         --step drives it through MpmPipeline::step (adaptive frames of FRAME_DT) instead of fixed-dt batches
  slabs  [--world 4] [--scene column|cube] [--per-slab 10e6] [--substeps 1000] [--report 200] [--drift V] [--single] [--strong]
         K x-slabs of ONE device through sparkl_hip_group_substeps (device-copy transport of the loop that uses RCCL between ranks):
         per-slab counts, cuts, collective re-bins, validator, id completeness; --single also times one pipeline holding the whole
         column (the fixed cost of the slab loop: printed per slab-substep); --strong cuts the benchmark's own 10 M column --world ways instead
         of scaling it (the geometry of strong scaling: thin slabs, full cross-section); SPARKL_HIP_OVERLAP=0/1/cal picks the schedule,
         SPARKL_HIP_DIRECT=0 the staged exchange (flag reduce, pack, copies, merge) instead of the peer-direct one
  step   [--frames 60] [--frame-dt 0.0166667]
         MpmPipeline::step against the fixed-dt batch at the same step length and the same phase of the 10M collapse"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from sparkl_amd import _lib, scenes  # noqa: E402
from sparkl_amd.pipeline import MpmPipeline  # noqa: E402

RS_NAMES = ["nruns", "maxlen", "distinct", "chunks4", "rounds", "iters", "pcount/64", "live", "rows", "layers", "|S|<1e-3,3e-3,1e-2,3e-2,.1,.3,inf", "maxcell", "iters_multi"]


def scene_of(name):
    if name == "sand10m":
        return scenes.sand_column(200, 250, 200)
    if name == "neo1m":
        return scenes.elastic_cube(100)
    if name == "fluid20m":
        return scenes.fluid_dam(344, 240, 244)
    if name == "cube":
        return scenes.cube_through_sand(240, 60, 120)
    raise SystemExit(f"unknown scene {name}")


def spans(k):
    def avg(ms, n):
        return ms / max(n, 1)
    return (f"g2p2g {avg(k.g2p2g_ms, k.g2p2g_launches):.4f} p2g {avg(k.p2g_ms, k.p2g_launches):.4f} g2p {avg(k.g2p_ms, k.g2p_launches):.4f} "
            f"grid {avg(k.grid_update_ms, k.grid_update_launches):.4f} sort {k.sort_ms:.1f} reorder {k.reorder_ms:.2f} ms, re-bins {int(k.resorts)}, no-op launches {int(k.noop_launches)}")


def run_stats(p):
    out = (C.c_ulonglong * (13 * 65))()
    _lib.check(_lib.lib().sparkl_hip_debug_run_stats(p.h, out, C.c_uint64(len(out))))
    a = np.array(out, dtype=np.float64).reshape(13, 65)
    for k, name in enumerate(RS_NAMES):
        h = a[k]
        tot = h.sum()
        if tot == 0:
            continue
        cum = np.cumsum(h) / tot
        q = [int(np.searchsorted(cum, x)) for x in (0.1, 0.5, 0.9, 0.99)]
        top = " ".join(f"{i}:{h[i] / tot:.3f}" for i in np.argsort(-h)[:8] if h[i] > 0)
        print(f"  {name:10s} n {int(tot):8d} mean {(h * np.arange(65)).sum() / tot:6.2f} p10/50/90/99 {q}  top {top}")
    st = a[10]   # waves by wave maximum, thresholds 1e-3 3e-3 1e-2 3e-2 0.1 0.3 inf: |F^T F - I| (bins 0..6), |log_vol_gain| (16..22), the larger (32..38)
    n = max(st[:7].sum(), 1)
    for off, name in ((0, "|S|"), (16, "|lvg|"), (32, "max")):
        print(f"  waves by {name:6s}", " ".join(f"{x / n:.3f}" for x in st[off:off + 7]))


def cmd_phase(a):
    sc = scene_of(a.config)
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        if a.reorder_heads is not None:
            p.reorder_policy_set(a.reorder_heads * len(sc.particles), 0.0)
        done = 0
        for m in a.marks:
            if m > done:
                p.substeps(sc.params, sc.gravity, sc.fixed_dt, m - done)
                done = m
            p.sync()
            print(f"== after {done} substeps: particle blocks {p.num_particle_blocks()}, node blocks {p.num_active_blocks()}", flush=True)
            if a.stats:
                run_stats(p)
            p.clock_enable(True, 2)
            p.clock_read(True)
            p.sync()
            t0 = time.perf_counter()
            p.substeps(sc.params, sc.gravity, sc.fixed_dt, a.steps)
            p.sync()
            ms = 1e3 * (time.perf_counter() - t0) / a.steps
            done += a.steps
            k = p.clock_read(True)
            p.clock_enable(False)
            print(f"  {a.steps} substeps: {ms:.4f} ms per substep; {spans(k)}, block reorders so far {p.reorder_policy_stats()['reorders']}", flush=True)
            if a.stats:
                print("  after them:")
                run_stats(p)


def cmd_soak(a):
    sc = scene_of(a.config)
    if a.step:
        sc.params.dt = a.step
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        if a.reorder_heads is not None:
            p.reorder_policy_set(a.reorder_heads * len(sc.particles), 0.0)
        p.clock_enable(True, stride=8)
        for chunk in range(a.chunks):
            p.clock_read(reset=True)
            p.sync()
            t0 = time.perf_counter()
            nsub = a.chunk
            if a.step:
                nsub = 0
                for _ in range(a.chunk):
                    nsub += p.step(sc.params, sc.gravity, timings=True)[1].num_substeps
            else:
                p.substeps(sc.params, sc.gravity, sc.fixed_dt, a.chunk)
            p.sync()
            ms = 1e3 * (time.perf_counter() - t0) / max(nsub, 1)
            k = p.clock_read()
            print(f"chunk {chunk}: {nsub} substeps, {ms:.4f} ms per substep ({len(sc.particles) / ms / 1e6:.2f} G particle-substeps/s); {spans(k)}; "
                  f"blocks {p.num_particle_blocks()} / {p.num_active_blocks()}, reorders {p.reorder_policy_stats()['reorders']}, step stats {p.step_stats()}", flush=True)
        c = p.grid_check()
        part = p.read_particles()
        pos, v = part.position["point"], part.velocity["vector"]
        print("grid check:", c)
        print("finite:", bool(np.isfinite(pos).all() and np.isfinite(v).all()), "failed:", int(part.status["failed"].sum()), "vmax %.2f" % float(np.abs(v).max()))


def cmd_slabs(a):
    import torch
    from sparkl_amd.particles import ParticleComponents
    from sparkl_amd.slab import INT32_MAX, INT32_MIN, block_cuts_balanced
    world = a.world
    stream = torch.cuda.Stream()
    pipes = []
    if a.scene == "cube":
        sc = scenes.cube_through_sand(240, 60, 120)
        if a.drift:
            sc.particles.velocity["vector"][:, 0] += np.float32(a.drift)
        h = sc.cell_width
        cell = np.round(sc.particles.position["point"][:, 0] / np.float32(h)).astype(np.int64) - 1
        cuts = block_cuts_balanced(cell >> 2, world)
        gids = [np.nonzero(((cell >> 2) >= cuts[r]) & ((cell >> 2) < cuts[r + 1]))[0] for r in range(world)]
        for r in range(world):
            comps = ParticleComponents(0)
            for k in ParticleComponents.__slots__:
                setattr(comps, k, np.ascontiguousarray(getattr(sc.particles, k)[gids[r]]))
            p = MpmPipeline(h, device_id=0, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
            p.set_slab(0, cuts[r], cuts[r + 1], id_offset=int(sum(len(g) for g in gids[:r])))
            p.set_models(sc.models)
            p.set_colliders(sc.colliders)
            p.upload_particles(comps)
            pipes.append(p)
        total, params, gravity, dt = len(sc.particles), sc.params, sc.gravity, sc.fixed_dt
    else:
        f = (a.per_slab / 10e6) ** (1 / 3)
        nx, ny, nz = max(8, int(round(200 * f / 8)) * 8), int(round(250 * f)), int(round(200 * f))
        if a.strong:   # the 10 M column itself, cut along x at block boundaries (8 particles): 200 = 8 x 25 -> slabs of 24 or 32 particles are not equal; use 192 = 8 x 24
            nx, ny, nz = max(8, (192 // world) // 8 * 8), 250, 200
        single_ms = None
        if a.single:
            sc = scenes.sand_column(world * nx, ny, nz)
            with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream) as p:
                p.load_scene(sc)
                p.substeps(sc.params, sc.gravity, sc.fixed_dt, 20)
                p.sync()
                t0 = time.perf_counter()
                p.substeps(sc.params, sc.gravity, sc.fixed_dt, a.report)
                p.sync()
                single_ms = 1e3 * (time.perf_counter() - t0) / a.report
                print(f"one pipeline, {len(sc.particles)} particles: {single_ms:.4f} ms per substep", flush=True)
            del sc
        total = 0
        for r in range(world):
            s = scenes.sand_column(nx, ny, nz, x_slab=(r, world))
            blocks = nx // 8
            p = MpmPipeline(s.cell_width, device_id=0, domain_log2_bytes=s.domain_log2_bytes, stream=stream.cuda_stream)
            p.set_slab(0, INT32_MIN if r == 0 else 1 + r * blocks, INT32_MAX if r == world - 1 else 1 + (r + 1) * blocks, id_offset=r * len(s.particles))
            p.set_models(s.models)
            p.set_colliders(s.colliders)
            p.upload_particles(s.particles)
            pipes.append(p)
            total += len(s.particles)
            params, gravity, dt = s.params, s.gravity, s.fixed_dt
    print(total, "particles on", world, "slabs:", [len(p) for p in pipes], flush=True)
    if a.scene == "cube":
        single_ms = None
    pipes[0].clock_enable(True, 4)
    for block in range(a.substeps // a.report):
        pipes[0].clock_read(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        MpmPipeline.group_substeps(pipes, params, gravity, dt, a.report)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / a.report
        n = [len(p) for p in pipes]
        chk = [p.grid_check() for p in pipes]
        bad = [sum(c[k] for k in ("blocks_not_in_hash", "particles_outside_their_block", "missing_neighbour_blocks", "coverage_errors")) for c in chk]
        k = pipes[0].clock_read(True)
        print(f"substeps {a.report * block:4d}..: {ms:.4f} ms per substep ({total / ms / 1e6:.2f} G particle-substeps/s), per slab {n} max/mean {max(n) * world / sum(n):.3f}, "
              f"re-bins {[p.group_rebins() for p in pipes]}, cuts {[p.slab_get()[1:3] for p in pipes]}, overlap {pipes[0].overlap_info()}, "
              f"member 0 halo {k.halo_ms / max(k.sampled_substeps, 1):.4f} ms per substep, validator violations {bad}, direct {pipes[0].direct_info()}"
              + (f", FIXED COST {1e3 * (ms - single_ms) / world:.1f} us per slab-substep over the single pipeline" if single_ms else ""), flush=True)
    ids = np.concatenate([p.read_sorted_particle_ids() for p in pipes])
    print("ids unique and complete:", bool(np.array_equal(np.sort(ids), np.arange(total))))
    for p in pipes:
        p.close()


def cmd_step(a):
    sc = scenes.sand_column(200, 250, 200)
    sc.params.dt = a.frame_dt
    res = {}
    for mode in ("step", "fixed"):
        with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
            p.load_scene(sc)
            out = []
            for _ in range(a.frames):
                p.sync()
                t0 = time.perf_counter()
                if mode == "step":
                    n = p.step(sc.params, sc.gravity, timings=True)[1].num_substeps
                else:
                    n = 6
                    p.substeps(sc.params, sc.gravity, sc.params.dt / 6.0, 6)
                p.sync()
                out.append(1e3 * (time.perf_counter() - t0) / n)
            res[mode] = out
            print(mode, "ms per substep, first / middle / last ten frames:", " / ".join(f"{sum(out[s:s + 10]) / 10:.3f}" for s in (0, a.frames // 2, a.frames - 10)),
                  p.step_stats() if mode == "step" else "", flush=True)
    for s in (0, a.frames // 2, a.frames - 10):
        print(f"frames {s}-{s + 9}: step / fixed = {sum(res['step'][s:s + 10]) / sum(res['fixed'][s:s + 10]):.3f}")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("phase")
    s.add_argument("--config", default="sand10m")
    s.add_argument("--marks", type=int, nargs="*", default=[100, 640, 2000])
    s.add_argument("--steps", type=int, default=128)
    s.add_argument("--stats", action="store_true")
    s.add_argument("--reorder-heads", type=float, default=None)
    s.set_defaults(fn=cmd_phase)
    s = sub.add_parser("soak")
    s.add_argument("--config", default="sand10m")
    s.add_argument("--chunks", type=int, default=8)
    s.add_argument("--chunk", type=int, default=500)
    s.add_argument("--step", type=float, default=0.0)
    s.add_argument("--reorder-heads", type=float, default=None)
    s.set_defaults(fn=cmd_soak)
    s = sub.add_parser("slabs")
    s.add_argument("--world", type=int, default=4)
    s.add_argument("--scene", choices=("column", "cube"), default="cube")
    s.add_argument("--per-slab", type=float, default=10e6)
    s.add_argument("--substeps", type=int, default=1000)
    s.add_argument("--report", type=int, default=200)
    s.add_argument("--drift", type=float, default=0.0)
    s.add_argument("--single", action="store_true")
    s.add_argument("--strong", action="store_true")
    s.set_defaults(fn=cmd_slabs)
    s = sub.add_parser("step")
    s.add_argument("--frames", type=int, default=60)
    s.add_argument("--frame-dt", type=float, default=1.0 / 60.0)
    s.set_defaults(fn=cmd_step)
    a = ap.parse_args()
    a.fn(a)


if __name__ == "__main__":
    main()
