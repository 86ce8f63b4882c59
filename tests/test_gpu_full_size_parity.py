"""Oracle parity at BASELINE.json's full sizes (VERDICT r02 "missing" #6): the binning, hash and tile paths of the real
configurations are held to the CHECKER (tests/parity.py::parity_verdict -- f64 arbiter, f32 oracle bounded beside it), not only to
size-independent properties (tests/test_gpu_full_size.py).  The oracle does ~2.7 M particle-substeps/s on 16-32 threads, so the
substep counts are what it finishes in about a minute per case.  Position and velocity bounds: the plain 1e-4 against the f64 arbiter
(round 5: positions are held relative to the home block, so the f32 world-coordinate drift -- 3.9e-4 h for the f32 oracle on the 10.8 m
cube over 100 substeps -- is no longer the HIP path's).
  configs[1]  1 M neo-Hookean cube       x 100 substeps
  configs[2] 10 M Drucker-Prager column  x 100 substeps
  a 1 M-particle FLOWING Drucker-Prager column (sheared from the start) x 100 substeps: in-place reorders, re-bins, grouped (unsorted) waves,
              both series tiers and the eigenbasis path of the return map meet the oracle at size
  configs[3]  2 M-particle slice of the 20 M fluid dam (64 ppc, volume recompute sweep) x 100 substeps (after 10 the dam has
              barely started to move: max |v| = 0.02 m/s, and eps_v -- relative to it -- only measures the f32 noise of the stiff
              equation of state, 3.5e-4 for the HIP path and 4.2e-4 for the f32 oracle alike)"""
import json
import os
import time

import numpy as np
import pytest

from sparkl_amd import scenes
from tests.parity import parity_verdict

pytestmark = pytest.mark.gpu
VIOLATIONS = ("bad_particle_ids", "blocks_not_in_hash", "particles_outside_their_block", "missing_neighbour_blocks", "coverage_errors")


def _threads():
    return max(1, min(32, len(os.sched_getaffinity(0))))


def _oracle_run(sc, nsub, f64, threads=None):
    from oracle import Oracle
    o = Oracle(sc.cell_width, f64=f64, domain_log2_bytes=sc.domain_log2_bytes, threads=threads or _threads())
    o.load_scene(sc)
    t0 = time.time()
    for _ in range(nsub):
        o.substep_fixed(sc.params, sc.gravity, sc.fixed_dt)
    dt = time.time() - t0
    out = o.get_particles()
    o.close()
    return out, dt


def _oracle_pair(sc, nsub):
    """The f32 and the f64 oracle side by side, half of the host's threads each (ctypes releases the GIL; the oracle stops scaling at ~32
    threads anyway): (r32, t32), (r64, t64)."""
    import concurrent.futures as cf
    half = max(1, min(32, len(os.sched_getaffinity(0)) // 2))
    with cf.ThreadPoolExecutor(2) as ex:
        f32 = ex.submit(_oracle_run, sc, nsub, False, half)
        f64 = ex.submit(_oracle_run, sc, nsub, True, half)
        return f32.result(), f64.result()


def _case(name, sc, nsub, arbiter="f64"):
    from sparkl_amd.pipeline import MpmPipeline
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, nsub)
        got = p.read_particles()
        chk = p.grid_check()
        npb = p.num_particle_blocks()
    assert all(int(chk[k]) == 0 for k in VIOLATIONS), chk
    (r32, t32), (r64, t64) = _oracle_pair(sc, nsub)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width, arbiter=arbiter)
    rec = {"test": name, "particles": len(sc.particles), "substeps": nsub, "particle_blocks": npb, "oracle_seconds": [t32, t64], "threads": _threads(), **rep}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "parity.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(name, json.dumps({k: rec[k] for k in ("particles", "substeps", "oracle_seconds")}), "vs f64", rep["vs_f64"], "| vs f32", rep["vs_f32"], "| f32 vs f64", rep["oracle_f32_vs_f64"])
    assert not bad, (bad, rep)
    return got


def test_config2_neo_hookean_1M_100_substeps_vs_oracle():
    sc = scenes.baseline_config(1)
    assert len(sc.particles) == 1_000_000
    _case("full_size_neo1m_k100", sc, 100)


def test_config3_drucker_prager_10M_100_substeps_vs_oracle():
    """BASELINE configs[2] over the 100 substeps north_star names (rounds 2 - 4 held it for 30: the oracle's serial sort made 100 take half an
    hour; with the stable parallel merge sort and both oracles side by side it is ~6 minutes on the GPU box's host)."""
    sc = scenes.baseline_config(2)
    assert len(sc.particles) == 10_000_000
    _case("full_size_sand10m_k100", sc, 100)


def test_flowing_sand_1M_100_substeps_vs_oracle():
    """The flowing phase at size: a 100^3 Drucker-Prager column with a shear rate that grows along x (up to 3 1/s: the top right corner
    moves ~4 cells in the 100 substeps; the in-place reorder is armed at 0.05 excess heads per particle instead of the default 0.2 so that it
    acts between the re-bins of so short a run).  Asserted from the library's own statistics that the paths the lattice never takes DID run --
    at least one re-bin, one in-place block reorder, waves whose lanes are no longer sorted by cell (the grouped scatter), waves on
    the quartic tier and on the second tier of the series return map -- and the state is held to the same gate as everywhere."""
    import ctypes as C
    from sparkl_amd import _lib
    from sparkl_amd.pipeline import MpmPipeline
    sc = scenes.sand_column(100, 100, 100, with_block=False)
    assert len(sc.particles) == 1_000_000
    pos = sc.particles.position["point"]
    rel = (pos - pos.min(0)) / np.ptp(pos, axis=0)
    rate = 3.0 * rel[:, 0]
    v = sc.particles.velocity["vector"]
    v[:, 0] = rate * (pos[:, 1] - pos[:, 1].min())
    v[:, 1] = -0.3 * rate * (pos[:, 1] - pos[:, 1].min())
    nsub = 100
    strain_bins, unsorted_waves = np.zeros(7), 0.0
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.reorder_policy_set(0.05 * len(sc.particles), 0.0)
        done = 0
        for k in (25, 50, 75, 100):
            p.substeps(sc.params, sc.gravity, sc.fixed_dt, k - done)
            done = k
            out = (C.c_ulonglong * (13 * 65))()
            _lib.check(_lib.lib().sparkl_hip_debug_run_stats(p.h, out, C.c_uint64(len(out))))
            hist = np.array(out, dtype=np.float64).reshape(13, 65)
            strain_bins += hist[10][:7]          # waves by max |F^T F - I|: < 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, inf
            runs, cells = (hist[0] * np.arange(65)).sum(), (hist[2] * np.arange(65)).sum()
            unsorted_waves += runs - cells       # a wave whose lanes are sorted by cell has as many lane runs as distinct cells
        got = p.read_particles()
        chk = p.grid_check()
        clock = p.clock_read(reset=False)
        pol = p.reorder_policy_stats()
    assert all(int(chk[k]) == 0 for k in VIOLATIONS), chk
    print("flowing sand 1M: re-bins", int(clock.resorts), "block reorders", pol["reorders"], "excess lane runs", unsorted_waves, "waves by strain", strain_bins)
    assert clock.resorts >= 2, "no re-bin beyond the initial one"
    assert pol["reorders"] >= 1 and unsorted_waves > 0
    assert strain_bins[:4].sum() > 0 and strain_bins[4] > 0, strain_bins     # both series tiers (the eigenbasis path beyond them: tests/test_gpu_parity.py's sheared column)
    (r32, t32), (r64, t64) = _oracle_pair(sc, nsub)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    rec = {"test": "flowing_sand1m_k100", "particles": len(sc.particles), "substeps": nsub, "oracle_seconds": [t32, t64], "threads": _threads(),
           "rebins": int(clock.resorts), "block_reorders": int(pol["reorders"]), **rep}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "parity.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("flowing_sand1m_k100 vs f64", rep["vs_f64"], "| vs f32", rep["vs_f32"], "| f32 vs f64", rep["oracle_f32_vs_f64"])
    assert not bad, (bad, rep)


def test_config4_fluid_2M_slice_100_substeps_vs_oracle():
    """160 x 112 x 112 = 2.0 M particles of the dam (64 per cell): mass-only sweep, volume gather, hash of a 64 ppc cloud."""
    sc = scenes.fluid_dam(160, 112, 112, name="config4_fluid_slice")
    assert sc.params.force_fluids_volume_recomputation and len(sc.particles) == 160 * 112 * 112
    # the f32 reference itself is 4.6e-3 from its f64 trajectory here (pressure noise of the stiff equation of state): held to the
    # f32 oracle, see parity_verdict(arbiter="either")
    _case("full_size_fluid2m_k100", sc, 100, arbiter="either")
