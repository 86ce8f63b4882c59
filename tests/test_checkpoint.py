"""Serialisation / checkpoint of particle + model state: sparkl_amd/checkpoint.py."""
import struct

import numpy as np
import pytest

from sparkl_amd import abi, checkpoint, scenes
from sparkl_amd.models import (CorotatedLinearElasticity, DruckerPragerPlasticity, MaximumStressFailure, MonaghanSphEos, NaccPlasticity,
                               NeoHookeanElasticity, ParticleModel, ParticleModelSet, RankinePlasticity, SnowPlasticity)
from sparkl_amd.particles import ParticleComponents


def _all_models():
    ms = ParticleModelSet()
    ms.insert(ParticleModel.with_plasticity(CorotatedLinearElasticity(1.0e7, 0.2), DruckerPragerPlasticity(1.0e7, 0.2)))
    ms.insert(ParticleModel.new(NeoHookeanElasticity(1.0e6, 0.3)))
    ms.insert(ParticleModel.new(MonaghanSphEos(1.0e6, 7, 1.01e-3)))
    ms.insert(ParticleModel.with_plasticity(CorotatedLinearElasticity(1.4e5, 0.2), SnowPlasticity()))
    ms.insert(ParticleModel.with_plasticity(NeoHookeanElasticity(1.0e6, 0.3), RankinePlasticity(1.0e6, 0.3, 5.0e4, 5.0)))
    ms.insert(ParticleModel.with_plasticity(NeoHookeanElasticity(2.0e5, 0.3), NaccPlasticity(2.0e5, 0.3, 0.05, True, 1.5, 0.6)))
    ms.insert(ParticleModel.with_failure(CorotatedLinearElasticity(1.0e6, 0.3), MaximumStressFailure(5.0e4, 4.0e4)))
    return ms


def _some_particles():
    sc = scenes.sand_column(6, 5, 4, with_block=True)
    c = sc.particles
    rng = np.random.default_rng(5)
    n = len(c)
    c.velocity["vector"] = rng.normal(0, 1, (n, 3)).astype(np.float32)
    c.velocity_gradient["m"] = rng.normal(0, 1, (n, 9)).astype(np.float32)
    c.volume["deformation_gradient"] += rng.normal(0, 0.01, (n, 9)).astype(np.float32)
    c.status["model_index"] = rng.integers(0, 7, n)
    c.status["kinematic_vel_enabled"][::7] = 1
    c.status["kinematic_vel"][::7] = [1.0, 2.0, 3.0]
    c.status["is_static"][3::11] = 1
    c.status["failed"][5::13] = 1
    c.phase["psi_pos"] = rng.random(n).astype(np.float32)
    c.phase["phase"][::5] = 0.0
    c.data["log_vol_gain"] = rng.normal(0, 0.1, n).astype(np.float32)
    return c


def _same_models(a, b, n):
    sz = abi.C.sizeof(abi.ParticleModelC)
    for k in range(n):
        ma, mb = a[k], b[k]
        # compare through the serde bytes: union padding in the raw structs is not significant
        wa, wb = __import__("io").BytesIO(), __import__("io").BytesIO()
        checkpoint._model_bytes(wa, ma); checkpoint._model_bytes(wb, mb)
        assert wa.getvalue() == wb.getvalue(), k
    assert sz == 76


def test_native_checkpoint_round_trip(tmp_path):
    c, ms = _some_particles(), _all_models()
    params = abi.SolverParameters()
    params.max_num_substeps = 77
    path = tmp_path / "ck.npz"
    checkpoint.save(path, c, ms, 0.2, params=params, simulated_time=1.25, extra={"frame": 12})
    c2, ms2, header, params2 = checkpoint.load(path)
    for k in ParticleComponents.__slots__:
        assert getattr(c, k).tobytes() == getattr(c2, k).tobytes(), k
    assert len(ms2) == len(ms) == 7
    assert bytes(ms2.to_c_array()) == bytes(ms.to_c_array())
    assert header["cell_width"] == pytest.approx(0.2) and header["simulated_time"] == 1.25 and header["extra"] == {"frame": 12}
    assert params2.max_num_substeps == 77 and bytes(params2) == bytes(params)


def test_bincode_layout_round_trip_and_sizes():
    c, ms = _some_particles(), _all_models()
    pb, mb = checkpoint.dump_bincode(c, ms)
    # Vec<Particle>: u64 count, then per particle 236 B (+ 20 B when kinematic_vel is Some)
    n, nkin = len(c), int(c.status["kinematic_vel_enabled"].sum())
    per = 8 + 12 + 20 + 20 + 1 + 44 + 44 + 4 + 8 + 1 + 1 + 20 + 20 + 4 + 12 + 20 + 16 + 8 + 4
    assert struct.unpack_from("<Q", pb)[0] == n and len(pb) == 8 + per * n + 20 * nkin
    # first particle: arena index + generation, mass
    assert struct.unpack_from("<II", pb, 8) == (int(c.status["model_index"][0]), 0)
    assert struct.unpack_from("<f", pb, 16)[0] == c.volume["mass"][0]
    # Vec<ParticleModel>: model 0 = CorotatedLinearElasticity (variant 0) + Some(DruckerPrager (variant 0)) + None
    assert struct.unpack_from("<Q", mb)[0] == 7
    assert struct.unpack_from("<I", mb, 8)[0] == 0 and mb[8 + 4 + 13] == 1 and struct.unpack_from("<I", mb, 8 + 4 + 13 + 1)[0] == 0
    c2, ms2 = checkpoint.load_bincode(pb, mb)
    for k in ("position", "velocity", "velocity_gradient", "data"):
        assert getattr(c, k).tobytes() == getattr(c2, k).tobytes(), k
    for f in ("mass", "volume0", "radius0", "deformation_gradient", "plastic_deformation_gradient_det"):
        np.testing.assert_array_equal(c.volume[f], c2.volume[f])
    for f in ("failed", "is_static", "kinematic_vel_enabled", "model_index"):
        np.testing.assert_array_equal(c.status[f], c2.status[f])
    np.testing.assert_array_equal(c.status["kinematic_vel"][c.status["kinematic_vel_enabled"] == 1], c2.status["kinematic_vel"][c2.status["kinematic_vel_enabled"] == 1])
    np.testing.assert_array_equal(c.phase["phase"], c2.phase["phase"])
    np.testing.assert_array_equal(c.phase["psi_pos"], c2.phase["psi_pos"])
    _same_models(ms.to_c_array(), ms2.to_c_array(), 7)
    assert checkpoint.dump_bincode(c2, ms2) == (pb, mb)
    with pytest.raises(ValueError):
        checkpoint.load_bincode(pb + b"\\0", mb)


def test_version_1_checkpoints_still_load(tmp_path):
    """NATIVE_VERSION 2 added the ParticleFracture component; a version-1 file (no component) must keep loading."""
    import json
    sc = scenes.sand_column(4, 4, 4, with_block=False)
    path = tmp_path / "v2.npz"
    checkpoint.save(path, sc.particles, sc.models, sc.cell_width)
    z = dict(np.load(path, allow_pickle=False))
    header = json.loads(bytes(z["header"]).decode())
    assert header["version"] == checkpoint.NATIVE_VERSION == 2
    header["version"] = 1
    header.pop("has_fracture", None)
    z["header"] = np.frombuffer(json.dumps(header).encode(), dtype=np.uint8)
    old = tmp_path / "v1.npz"
    np.savez(old, **z)
    c, models, h1, params = checkpoint.load(old)
    assert h1["version"] == 1 and h1["fracture"] is None and len(c) == len(sc.particles)


def test_native_checkpoint_carries_the_fracture_component(tmp_path):
    """ParticleFracture (crack_propagation_factor, crack_threshold) rides in the native checkpoint; a ModifiedEigenerosion checkpoint
    without it is refused (it would resume with the crack sweep silently off)."""
    rng = np.random.default_rng(4)
    sc = scenes.sand_column(6, 5, 6, with_block=False)
    n = len(sc.particles)
    frac = np.zeros(n, dtype=abi.FRACTURE_DTYPE)
    frac["crack_propagation_factor"] = rng.random(n).astype(np.float32)
    frac["crack_threshold"] = rng.random(n).astype(np.float32)
    params = abi.SolverParameters.from_buffer_copy(bytes(sc.params))
    params.damage_model = abi.DAMAGE_MODIFIED_EIGENEROSION
    with pytest.raises(ValueError, match="ModifiedEigenerosion"):
        checkpoint.save(tmp_path / "bad.npz", sc.particles, sc.models, sc.cell_width, params=params)
    with pytest.raises(ValueError, match="fracture records"):
        checkpoint.save(tmp_path / "bad.npz", sc.particles, sc.models, sc.cell_width, params=params, fracture=frac[:-1])
    checkpoint.save(tmp_path / "ok.npz", sc.particles, sc.models, sc.cell_width, params=params, fracture=frac)
    c, models, header, p2 = checkpoint.load(tmp_path / "ok.npz")
    assert header["has_fracture"] and p2.damage_model == abi.DAMAGE_MODIFIED_EIGENEROSION
    np.testing.assert_array_equal(header["fracture"], frac)
    checkpoint.save(tmp_path / "plain.npz", sc.particles, sc.models, sc.cell_width, params=sc.params)
    assert checkpoint.load(tmp_path / "plain.npz")[2]["fracture"] is None


@pytest.mark.gpu
def test_resume_under_modified_eigenerosion_keeps_cracking(tmp_path):
    """A ModifiedEigenerosion run checkpointed half way (checkpoint.save_pipeline / resume) ends where the uninterrupted run ends:
    same cracked set (threshold decisions on the line may flip), same particle state to f32 rounding."""
    from sparkl_amd.models import CorotatedLinearElasticity
    from sparkl_amd.pipeline import MpmPipeline
    from tests.parity import particle_errors
    sc = scenes.block_on_floor(12, CorotatedLinearElasticity(1.0e6, 0.3), expand=6.0, v0=(0.0, 0.0, 0.0), gravity=(0.0, 0.0, 0.0), name="eigenerosion")
    dt, n = sc.fixed_dt, len(sc.particles)
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, 60)
        psi60 = p.read_particles().phase["psi_pos"]
    frac = np.zeros(n, dtype=abi.FRACTURE_DTYPE)
    frac["crack_propagation_factor"] = 1.0
    frac["crack_threshold"] = np.float32(sc.cell_width * np.percentile(psi60, 70))
    sc.params.damage_model = abi.DAMAGE_MODIFIED_EIGENEROSION
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.set_fracture(frac)
        p.substeps(sc.params, sc.gravity, dt, 100)
        ref = p.read_particles()
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.set_fracture(frac)
        p.substeps(sc.params, sc.gravity, dt, 50)
        checkpoint.save_pipeline(tmp_path / "mid.npz", p, sc.models, params=sc.params, simulated_time=50 * dt)
    with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        header, params = checkpoint.resume(p, tmp_path / "mid.npz")
        assert header["has_fracture"] and params.damage_model == abi.DAMAGE_MODIFIED_EIGENEROSION
        p.set_colliders(sc.colliders)
        p.substeps(params, sc.gravity, dt, 50)
        got = p.read_particles()
        np.testing.assert_array_equal(p.read_fracture(), frac)
    cracked_ref, cracked = ref.phase["phase"] == 0.0, got.phase["phase"] == 0.0
    assert n // 20 < int(cracked_ref.sum()) < n - n // 20, "the crack front must pass through part of the block"
    assert int((cracked != cracked_ref).sum()) <= max(2, int(cracked_ref.sum()) // 100)
    c_mid = checkpoint.load(tmp_path / "mid.npz")[0]
    assert int(cracked.sum()) > int((c_mid.phase["phase"] == 0.0).sum()), "cracking must go on after the resume"
    e = particle_errors(got, ref, sc.cell_width)
    assert e["eps_x"] < 1e-4 and e["eps_v"] < 1e-3, e


@pytest.mark.gpu
def test_resume_from_checkpoint_matches_uninterrupted_run(tmp_path):
    from sparkl_amd.pipeline import MpmPipeline
    from tests.parity import particle_errors
    sc = scenes.sand_column(16, 10, 12, with_block=True)
    dt = 2.5e-4
    with MpmPipeline(sc.cell_width) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, 80)
        ref = p.read_particles()
    with MpmPipeline(sc.cell_width) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, 40)
        checkpoint.save(tmp_path / "mid.npz", p.read_particles(), sc.models, sc.cell_width, params=sc.params, simulated_time=40 * dt)
    c, models, header, params = checkpoint.load(tmp_path / "mid.npz")
    with MpmPipeline(header["cell_width"]) as p:
        p.set_models(models)
        p.set_colliders(sc.colliders)
        p.upload_particles(c)
        p.substeps(params, sc.gravity, dt, 40)
        got = p.read_particles()
    e = particle_errors(got, ref, sc.cell_width)
    # the re-bin after the upload orders particles differently from the uninterrupted run: f32 summation order only
    assert e["eps_x"] < 1e-5 and e["eps_v"] < 1e-4 and e["failed_mismatch"] == 0, e


@pytest.mark.gpu
def test_resume_from_checkpoint_matches_the_oracle(tmp_path):
    """Checkpoint taken from the HIP path after 40 substeps, resumed on the HIP path for 60 more: the end state is held to the
    parity gate against the ORACLE's uninterrupted 100 substeps (not against the HIP path itself), and the oracle resumed
    from the same checkpoint file must agree with it as well -- the file carries everything the dynamics need."""
    import os
    from oracle import Oracle
    from sparkl_amd.pipeline import MpmPipeline
    from tests.parity import parity_verdict
    sc = scenes.sand_column(16, 10, 12, with_block=True)
    dt = 2.5e-4
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    refs = []
    for f64 in (False, True):
        o = Oracle(sc.cell_width, f64=f64, domain_log2_bytes=sc.domain_log2_bytes, threads=threads)
        o.load_scene(sc)
        for _ in range(100):
            o.substep_fixed(sc.params, sc.gravity, dt)
        refs.append(o.get_particles())
        o.close()
    with MpmPipeline(sc.cell_width) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, 40)
        checkpoint.save(tmp_path / "mid.npz", p.read_particles(), sc.models, sc.cell_width, params=sc.params, simulated_time=40 * dt)
    c, models, header, params = checkpoint.load(tmp_path / "mid.npz")
    assert header["simulated_time"] == pytest.approx(40 * dt)
    with MpmPipeline(header["cell_width"]) as p:
        p.set_models(models)
        p.set_colliders(sc.colliders)
        p.upload_particles(c)
        p.substeps(params, sc.gravity, dt, 60)
        got = p.read_particles()
    rep, bad = parity_verdict(got, refs[0], refs[1], sc.cell_width)
    print("resume (HIP) vs uninterrupted oracle:", rep["vs_f64"], rep["oracle_f32_vs_f64"])
    assert not bad, (bad, rep)
    # the oracle resumed from the HIP checkpoint lands in the same place
    o = Oracle(header["cell_width"], domain_log2_bytes=sc.domain_log2_bytes, threads=threads)
    o.set_models(models)
    o.set_cuboids(sc.colliders)
    o.set_particles(c)
    for _ in range(60):
        o.substep_fixed(params, sc.gravity, dt)
    # (the f32 oracle is the reference's own arithmetic: it is held to ITS distance from the f64 trajectory, not to the 1e-4 of
    # the HIP gate, which it does not meet on this sand case)
    from tests.parity import TOL, particle_errors
    e2, fl = particle_errors(o.get_particles(), refs[1], sc.cell_width), particle_errors(refs[0], refs[1], sc.cell_width)
    o.close()
    assert e2["eps_x"] <= max(TOL, fl["eps_x"]) and e2["eps_v"] <= max(TOL, fl["eps_v"]) and e2["failed_mismatch"] == 0, (e2, fl)
