"""The flags mirror of the fused kernel (kernels.h, ParticleArrays::fl; k_g2p2g<.., MSET = 4>): when every binned particle carries the same
(mass, volume0, elastic hardening) and the models are the Drucker-Prager view, the kernel reads a 4-byte flags stream and takes the three constants
from StepParams instead of reading the 16-byte k record.  Same arithmetic on the same numbers: results must be those of the k-record kernel
(SPARKL_HIP_KUNIFORM=0) up to the order in which the LDS atomics of a tile happen to add (two runs of ONE kernel differ by that much: an f64 sum that
lands on an f32 rounding boundary), through re-bins, in-place block reorders and flags that change on the way (particles leaving the domain) -- the
same set of dead particles in particular; and the mirror must stand down as soon as the particle set is not uniform."""
import numpy as np
import pytest

from sparkl_amd import _lib, scenes
from tests.parity import particle_errors

pytestmark = pytest.mark.gpu
VIOLATIONS = ("bad_particle_ids", "blocks_not_in_hash", "particles_outside_their_block", "missing_neighbour_blocks", "coverage_errors")


@pytest.fixture(scope="module")
def Pipeline():
    from sparkl_amd.pipeline import MpmPipeline
    return MpmPipeline


def _mirror_active(p):
    return int(_lib.lib().sparkl_hip_debug_flags_mirror(p.h))


def _flowing_column():
    sc = scenes.sand_column(20, 14, 12)
    sc.particles.velocity["vector"][:] = np.float32([-14.0, 0.0, 2.0])   # towards the low x face of the grid: a layer leaves the domain; fragments the lane runs
    return sc


def test_mirror_kernel_matches_the_k_record_kernel(Pipeline, monkeypatch):
    sc = _flowing_column()
    dt = sc.fixed_dt
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SPARKL_HIP_KUNIFORM", mode)
        with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
            p.load_scene(sc)
            p.clock_enable(True)
            p.reorder_policy_set(1.0, 0.0)          # eager: every check reorders, the mirror has to follow its particles
            p.substeps(sc.params, sc.gravity, dt, 150)
            assert _mirror_active(p) == int(mode), "the mirror must be in use exactly when it is switched on (uniform single-body sand)"
            p.substeps(sc.params, sc.gravity, dt, 150)
            chk = p.grid_check()
            assert all(int(chk[k]) == 0 for k in VIOLATIONS), chk
            out[mode] = (p.read_particles(), int(p.clock_read().resorts), p.reorder_policy_stats()["reorders"])
    a, b = out["1"][0], out["0"][0]
    failed = a.status["failed"] == 1
    assert out["1"][1] >= 3 and out["1"][2] >= 2, out["1"][1:]
    assert 50 <= failed.sum() < len(failed), f"{int(failed.sum())} particles left the domain: the scene must change flags on the way"
    # the same particles died -- except one that ends within rounding noise of the face it dies behind (x / h = 0.5) may do so a substep apart
    h = sc.cell_width
    near = (np.abs(a.position["point"][:, 0] / h - 0.5) < 1e-4) | (np.abs(b.position["point"][:, 0] / h - 0.5) < 1e-4)
    assert near.sum() <= 8 and (a.status[~near].tobytes() == b.status[~near].tobytes())
    for comps in (a, b):   # (taken out of the comparison below: a dead particle is frozen one advection apart)
        comps.position["point"][near] = 0.0; comps.velocity["vector"][near] = 0.0; comps.status["failed"][near] = 0
        comps.volume["deformation_gradient"][near] = 0.0; comps.velocity_gradient["m"][near] = 0.0
    e = particle_errors(a, b, sc.cell_width)
    print("mirror vs k record:", e)
    assert e["eps_x"] <= 1e-5 and e["eps_v"] <= 1e-5 and e["eps_F"] <= 1e-5 and e["failed_mismatch"] == 0, e
    # (re-bins and reorders are threshold events of a run whose last bits depend on the order of the tile atomics: the counts agree to within one or two)
    assert abs(out["1"][1] - out["0"][1]) <= 2 and abs(out["1"][2] - out["0"][2]) <= 2, (out["1"][1:], out["0"][1:])


def test_mirror_stands_down_for_a_non_uniform_particle_set(Pipeline):
    sc = scenes.sand_column(12, 8, 8)
    dt = sc.fixed_dt
    with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, 4)
        assert _mirror_active(p) == 1
        n0 = len(p)
        heavy = scenes.sand_column(4, 4, 4).particles
        heavy.position["point"] += np.float32([4.0, 0.0, 0.0])
        heavy.volume["mass"] *= np.float32(1.5)
        p.append_particles(heavy)
        p.substeps(sc.params, sc.gravity, dt, 4)              # re-bins: one body heavier than the other
        assert _mirror_active(p) == 0
        p.remove_range(n0, n0 + len(heavy))
        p.substeps(sc.params, sc.gravity, dt, 4)              # uniform again
        assert _mirror_active(p) == 1
        assert np.isfinite(p.read_positions()["point"]).all()
    # two bodies of different density from the start (config 1's sand on an elastic block is not even all Drucker-Prager)
    sc2 = scenes.sand_column(10, 6, 8)
    sc2.particles.volume["mass"][::2] *= np.float32(1.25)
    with Pipeline(sc2.cell_width, domain_log2_bytes=sc2.domain_log2_bytes) as p:
        p.load_scene(sc2)
        p.substeps(sc2.params, sc2.gravity, dt, 4)
        assert _mirror_active(p) == 0


def test_fluid_sweep_on_the_mirror_matches_the_k_record_kernels(Pipeline, monkeypatch):
    """The fused fluid sweep (k_vol_p2g / k_g2p_mass<.., KU>) of a plain-fluid scene reads the mirror too: 20 / 84 instead of 32 / 96 bytes per particle."""
    sc = scenes.fluid_dam(24, 16, 16)
    assert sc.params.force_fluids_volume_recomputation
    dt = sc.fixed_dt
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SPARKL_HIP_KUNIFORM", mode)
        with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
            p.load_scene(sc)
            p.substeps(sc.params, sc.gravity, dt, 60)
            assert _mirror_active(p) == int(mode)
            p.substeps(sc.params, sc.gravity, dt, 60)
            chk = p.grid_check()
            assert all(int(chk[k]) == 0 for k in VIOLATIONS), chk
            out[mode] = p.read_particles()
    a, b = out["1"], out["0"]
    assert a.status.tobytes() == b.status.tobytes()
    e = particle_errors(a, b, sc.cell_width)
    print("fluid, mirror vs k record:", e)
    # (the weakly compressible EOS amplifies the summation-order noise of the mass grid: two runs of one kernel differ by this much)
    assert e["eps_x"] <= 1e-5 and e["eps_v"] <= 1e-4 and e["failed_mismatch"] == 0, e


def test_mirror_stays_in_step_while_another_kernel_stores_the_flags(Pipeline, monkeypatch):
    """The mirror is handed to every G2P store whatever kernel runs: here the model table gets a second (unused) neo-Hookean entry for 120 substeps --
    the generic fused kernel takes over and particles die on its watch -- and loses it again: the Drucker-Prager view comes back on a mirror that must have
    followed.  Same sequence with the mirror off as the reference."""
    import copy
    sc = _flowing_column()
    dt = sc.fixed_dt
    two = copy.deepcopy(sc.models)
    two.insert(scenes.elastic_cube(2).models.get(0))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SPARKL_HIP_KUNIFORM", mode)
        with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
            p.load_scene(sc)
            p.substeps(sc.params, sc.gravity, dt, 140)
            dead0 = int((p.read_statuses()["failed"] == 1).sum())
            p.set_models(two)
            p.substeps(sc.params, sc.gravity, dt, 120)        # k_g2p2g<.., MSET = 0>: flags stored by the generic kernel
            dead1 = int((p.read_statuses()["failed"] == 1).sum())
            p.set_models(sc.models)
            p.substeps(sc.params, sc.gravity, dt, 60)
            assert _mirror_active(p) == int(mode)
            out[mode] = p.read_particles()
            assert dead1 > dead0, (dead0, dead1)               # flags changed while the other kernel ran
    a, b = out["1"], out["0"]
    h = sc.cell_width
    near = (np.abs(a.position["point"][:, 0] / h - 0.5) < 1e-4) | (np.abs(b.position["point"][:, 0] / h - 0.5) < 1e-4)
    assert near.sum() <= 8 and (a.status[~near].tobytes() == b.status[~near].tobytes())
    for comps in (a, b):
        comps.position["point"][near] = 0.0; comps.velocity["vector"][near] = 0.0; comps.status["failed"][near] = 0
        comps.volume["deformation_gradient"][near] = 0.0; comps.velocity_gradient["m"][near] = 0.0
    e = particle_errors(a, b, h)
    assert e["eps_x"] <= 1e-5 and e["eps_v"] <= 1e-5 and e["failed_mismatch"] == 0, e
