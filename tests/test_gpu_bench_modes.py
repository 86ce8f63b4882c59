"""bench.py's multi-rank modes and BASELINE configs[4] at its real size, on the ONE GPU of the test box:
  * `--gpus 2 --config cube80m` / `--scaling strong` through the gloo one-device hook (two ranks, one device): the N > 1 launcher,
    the slab scene builders and the JSON contract;
  * cube_through_sand3 at 80 M particles as 8 x-slabs of one device through sparkl_hip_group_substeps (the loop the 8-GPU run
    uses with RCCL): size-independent properties -- ids complete, nothing lost or failed, mass where it was, validator clean."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
VIOLATIONS = ("bad_particle_ids", "blocks_not_in_hash", "particles_outside_their_block", "missing_neighbour_blocks", "coverage_errors")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_two_ranks(extra, port, env_extra=None):
    env = dict(os.environ, SPARKL_BENCH_BACKEND="gloo", SPARKL_BENCH_ONE_DEVICE="1")
    if env_extra is not None:   # replaces the gloo hook: e.g. the library's own group loop over the host-staged transport
        env = dict(os.environ, SPARKL_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-copy-peak", "--clock-wake-ms", "0"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


def test_bench_two_ranks_cube80m_scene_is_one_fixed_job():
    o = _bench_two_ranks(["--config", "cube80m", "--particles", "2e5"], 29541)
    assert o["scaling"] == "strong" and o["n_gpus"] == 2 and o["ranks"] == 2
    cfg = o["config"]
    assert cfg["name"] == "cube80m" and "parity_tolerance" in cfg
    # 2e5 -> bed 128 x 29 x 54 + cube 3^3 (bench.strong_dims), whatever the rank count
    assert cfg["particles_total"] == 128 * 29 * 54 + 27, cfg
    assert cfg["particles_per_gpu"] == 64 * 29 * 54 + 27          # rank 0 owns the kinematic cube
    assert o["value"] == pytest.approx(cfg["particles_total"] * o["steps"] / (o["ms_per_step"] * 1e-3 * o["steps"]), rel=1e-6)
    assert o["halo_ms_per_step"] >= 0.0 and o["roofline"]["frac"] > 0.0
    assert "slab_overhead_us" in o and abs(o["slab_overhead_us"]) < 1e3 * o["ms_per_step"]
    assert "cube80m strong scaling (not BASELINE.json's headline metric)" in o["metric"]


def test_bench_two_ranks_strong_sand_keeps_the_total():
    o = _bench_two_ranks(["--scaling", "strong", "--particles", "2e5"], 29543)
    assert o["scaling"] == "strong" and o["ranks"] == 2
    assert o["config"]["particles_total"] == 2 * o["config"]["particles_per_gpu"] == 64 * 58 * 54, o["config"]
    assert o["metric"].startswith("particle-steps/sec (whole node) + achieved HBM GB/s, 3D sand 10M particles")
    # the DEFAULT is the same fixed job since round 6 (the metric's 10 M column, cut N ways), with the weak figure riding along
    d = _bench_two_ranks(["--particles", "2e5"], 29549)
    assert d["scaling"] == "strong" and d["config"]["particles_total"] == 64 * 58 * 54, d["config"]
    assert d["weak_scaling"]["scaling"] == "weak" and d["weak_scaling"]["particles_total"] == 2 * d["weak_scaling"]["particles_per_gpu"] and d["weak_scaling"]["value"] > 0
    w = _bench_two_ranks(["--scaling", "weak", "--particles", "1e5"], 29545)          # weak on request: one configured column per rank
    assert w["scaling"] == "weak" and w["config"]["particles_total"] == 2 * w["config"]["particles_per_gpu"] and "weak_scaling" not in w
    assert "weak scaling (not BASELINE.json's headline metric)" in w["metric"]


def test_bench_two_ranks_default_path_is_the_library_group_loop():
    """The DEFAULT N > 1 path of bench.py -- comm mode "rccl": sparkl_hip_comm_init from a broadcast unique id, then sparkl_hip_group_substeps, the C++
    loop with the flag all-reduce, the halo send / recv and the schedule calibration -- launched exactly as the driver launches it (torch.distributed.run,
    one process per rank), with the one substitution a one-GPU box needs: the host-staged transport behind the same function table
    (SPARKL_HIP_TRANSPORT=host; RCCL refuses two ranks on one device)."""
    o = _bench_two_ranks(["--particles", "4e5"], 29547, env_extra={"SPARKL_HIP_TRANSPORT": "host"})
    assert o["n_gpus"] == 2 and o["ranks"] == 2 and o["scaling"] == "strong"
    cfg = o["config"]
    assert cfg["particles_total"] == 2 * cfg["particles_per_gpu"]
    # round 6: between the two PROCESSES the halo and the re-bin flag travel peer-direct (hipIpc-mapped arenas), for the headline job and the weak one
    # between ranks the exchange is timed before the warmup (bench.py runs 112 untimed substeps for it and uploads the scene again): an answer and three numbers
    de = o["direct_exchange"]
    assert de["members"] == 2 and de["epochs"] >= 32 and de["chosen"] in ("direct", "staged") and de["timed_us_per_substep"]["direct"] > 0.0 and de["timed_us_per_substep"]["staged_in_line"] > 0.0, de
    assert o["weak_scaling"]["direct_exchange"]["chosen"] in ("direct", "staged"), o["weak_scaling"]
    assert o["value"] == pytest.approx(cfg["particles_total"] / (o["ms_per_step"] * 1e-3), rel=1e-6)
    assert o["roofline"]["frac"] > 0.0 and o["halo_ms_per_step"] >= 0.0
    assert o.get("comm") == "rccl" or "overlap" in o, sorted(o)


def test_config5_80M_as_eight_slabs_of_one_device():
    """cube_through_sand3 at its BASELINE size: 768 x 260 x 400 bed + 25^3 kinematic cube = 79.9 M particles in 8 x-slabs stepped by
    sparkl_hip_group_substeps (device-copy transport of the loop that uses RCCL between ranks): 20 substeps."""
    import torch
    from sparkl_amd import scenes
    from sparkl_amd.pipeline import MpmPipeline
    from sparkl_amd.slab import INT32_MAX, INT32_MIN
    import bench
    world = 8
    nx, ny, nz = bench.strong_dims(80.0e6, world, base=(800, 250, 400))
    assert (nx, ny, nz) == (768, 260, 400)
    stream = torch.cuda.Stream()
    pipes, n_in, mass_in = [], [], 0.0
    total = None
    for r in range(world):
        sc = scenes.cube_through_sand(nx // world, ny, nz, cube_n=25, x_slab=(r, world))
        total = sc.total_particles
        blocks = (nx // world) // 8
        lo = INT32_MIN if r == 0 else 1 + r * blocks
        hi = INT32_MAX if r == world - 1 else 1 + (r + 1) * blocks
        p = MpmPipeline(sc.cell_width, device_id=0, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
        p.set_slab(0, lo, hi, id_offset=sc.id_offset)
        p.set_models(sc.models)
        p.set_colliders(sc.colliders)
        p.upload_particles(sc.particles)
        n_in.append(len(sc.particles))
        mass_in += float(sc.particles.volume["mass"].astype(np.float64).sum())
        params, gravity, dt = sc.params, sc.gravity, sc.fixed_dt
        pipes.append(p)
        del sc
    assert sum(n_in) == total == 768 * 260 * 400 + 25 ** 3
    MpmPipeline.group_substeps(pipes, params, gravity, dt, 20)
    seen = np.zeros(total, dtype=bool)
    mass_out, n_out, failed, kin_x = 0.0, [], 0, []
    for p in pipes:
        p.sync()
        chk = p.grid_check()
        assert all(int(chk[k]) == 0 for k in VIOLATIONS), chk
        ids = p.read_sorted_particle_ids().astype(np.int64)
        assert not seen[ids].any(), "a particle lives on two slabs"
        seen[ids] = True
        n_out.append(len(p))
        mass_out += float(p.read_volumes()["mass"].astype(np.float64).sum())
        st = p.read_statuses()
        failed += int(st["failed"].sum())
        kin = st["kinematic_vel_enabled"] != 0
        if kin.any():
            kin_x.append(p.read_positions()["point"][kin, 0])
        vel = p.read_velocities()["vector"]
        assert np.isfinite(vel).all()
    for p in pipes:
        p.close()
    assert seen.all() and sum(n_out) == total, (int(seen.sum()), sum(n_out), total)
    assert failed == 0
    assert abs(mass_out - mass_in) <= 1e-9 * mass_in
    # the kinematic cube moved by 10 m/s * 20 dt and is still complete
    kin_x = np.concatenate(kin_x)
    assert len(kin_x) == 25 ** 3
    print("80M in 8 slabs: particles per slab", n_in, "->", n_out)
