"""`CudaColliderOptions` (src/cuda/cuda_collider_set.rs:22-38) through the C ABI (SparklColliderOptions, round 5) against the oracle, one
test per option, and the grid update's projection cache (src_kernels/cuda/grid_update.rs:111-146) against the uncached kernel."""
import copy
import os

import numpy as np
import pytest

from sparkl_amd import abi, scenes
from sparkl_amd._lib import SparklHipError
from tests.parity import parity_verdict

pytestmark = pytest.mark.gpu


@pytest.fixture
def Pipeline():
    from sparkl_amd.pipeline import MpmPipeline
    return MpmPipeline


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def _oracles(sc, nsteps, dt):
    from oracle import Oracle
    out = []
    for f64 in (False, True):
        o = Oracle(sc.cell_width, f64=f64, domain_log2_bytes=sc.domain_log2_bytes, threads=_threads())
        o.load_scene(sc)
        for _ in range(nsteps):
            o.substep_fixed(sc.params, sc.gravity, dt)
        out.append(o.get_particles())
        o.close()
    return out


def _hip(Pipeline, sc, nsteps, dt):
    with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, nsteps)
        return p.read_particles()


def _falling_sand():
    sc = scenes.sand_column(12, 6, 10, with_block=False)
    sc.particles.velocity["vector"][:, 1] = -2.0
    sc.particles.velocity["vector"][:, 0] = 1.5
    return sc


@pytest.mark.parametrize("handling", [abi.BOUNDARY_STICK, abi.BOUNDARY_NONE, abi.BOUNDARY_FRICTION, 2])
def test_per_collider_boundary_handling_overrides_the_solver_parameters(Pipeline, handling):
    """Two floors' worth of switch: SolverParameters says Friction; the collider's own grid_boundary_handling says Stick / None / Friction /
    FrictionZUp.  HIP path vs oracle under the gate, and the override must MATTER (Stick and None end elsewhere than the default)."""
    sc = _falling_sand()
    dt, n = 2.5e-4, 100
    base = _hip(Pipeline, sc, n, dt)
    sc.collider_options = {"cuboid": [abi.collider_options(grid_boundary_handling=handling)]}
    got = _hip(Pipeline, sc, n, dt)
    r32, r64 = _oracles(sc, n, dt)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    moved = np.abs(got.position["point"] - base.position["point"]).max() / sc.cell_width
    print("collider handling", handling, "vs f64", rep["vs_f64"]["eps_x"], rep["vs_f64"]["eps_v"], "| differs from the default by", moved, "h")
    assert not bad, (bad, rep)
    if handling in (abi.BOUNDARY_STICK, abi.BOUNDARY_NONE):
        assert moved > 1e-2, moved
    else:
        assert moved < 1e-6, moved     # Friction = the default; FrictionZUp on a floor whose normal points up (+y, z = 0 >= 0) is Friction too


def test_solver_parameters_none_with_one_collider_opting_in(Pipeline):
    """boundary_handling = None in the solver parameters used to skip the colliders altogether; a collider that carries its own handling
    still acts."""
    sc = _falling_sand()
    sc.params = copy.copy(sc.params)
    sc.params.boundary_handling = abi.BOUNDARY_NONE
    dt, n = 2.5e-4, 100
    free = _hip(Pipeline, sc, n, dt)
    sc.collider_options = {"cuboid": [abi.collider_options(grid_boundary_handling=abi.BOUNDARY_FRICTION)]}
    got = _hip(Pipeline, sc, n, dt)
    r32, r64 = _oracles(sc, n, dt)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    assert not bad, (bad, rep)
    assert np.abs(got.position["point"] - free.position["point"]).max() / sc.cell_width > 1e-2


@pytest.mark.parametrize("flip", [False, True])
def test_heightfield_flip_interior(Pipeline, flip):
    """gpu_collider.rs:81-89: with below_is_solid the solid side of a heightfield is below the surface, flip_interior puts it above.  Stick
    handling makes the difference visible: only nodes INSIDE are stopped."""
    # a sheet one cell thick, one cell above the valley bottom: every node it touches lies within the 4 h the grid update searches around a
    # node (the reference's GPU path looks 2 h far, its CPU path without limit: beyond the reach "inside" is simply never reported here)
    sc = scenes.sand_on_heightfield(16, 2, 12, below_is_solid=True)
    sc.particles.position["point"][:, 1] -= np.float32(sc.cell_width)
    sc.particles.velocity["vector"][:, 1] = -3.0
    sc.params = copy.copy(sc.params)
    sc.params.boundary_handling = abi.BOUNDARY_STICK
    sc.collider_options = {"heightfield": [abi.collider_options(flip_interior=flip)]}
    dt, n = 2.5e-4, 60
    got = _hip(Pipeline, sc, n, dt)
    r32, r64 = _oracles(sc, n, dt)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    vy = got.velocity["vector"][:, 1]
    print("heightfield flip_interior", flip, "vs f64", rep["vs_f64"]["eps_x"], rep["vs_f64"]["eps_v"], "mean vy", float(vy.mean()))
    assert not bad, (bad, rep)
    if flip:
        assert (np.abs(vy) < 1e-3).mean() > 0.5      # the column starts ABOVE the surface = inside the flipped solid: stuck at once
    else:
        assert vy.mean() < -0.2                        # falls (and is being stopped by the ground it has reached by now)


def test_penalty_stiffness_is_stored_and_ignored_like_the_reference(Pipeline):
    """src_kernels/cuda/particle_updater.rs:203-221 has the penalty force under `if false`: a collider's penalty_stiffness changes nothing in
    the reference, so by default it changes nothing here either -- same bits as without the option, and no detour through the RARE kernels."""
    sc = scenes.sand_column(8, 6, 8, with_block=False)
    sc.params = copy.copy(sc.params)
    sc.params.boundary_handling = abi.BOUNDARY_NONE
    sc.particles.position["point"][:, 1] -= np.float32(1.25 * sc.cell_width)
    dt, n = 2.5e-4, 80
    free = _hip(Pipeline, sc, n, dt)
    sc.collider_options = {"cuboid": [abi.collider_options(penalty_stiffness=2.0e5)]}
    got = _hip(Pipeline, sc, n, dt)
    for k in ("position", "velocity", "volume", "velocity_gradient"):
        assert getattr(got, k).tobytes() == getattr(free, k).tobytes(), k
    r32, r64 = _oracles(sc, n, dt)          # the oracle's default is the reference's too
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    assert not bad, (bad, rep)


def test_penalty_stiffness_pushes_particles_out_when_opted_in(Pipeline):
    """The INTENDED rule behind sparkl_hip_colliders_enable_penalty_force: a block that starts half inside a cuboid with a penalty stiffness and
    NO grid boundary handling is pushed out by f = (proj - x) k; without the stiffness it falls through.  HIP vs oracle (same switch) under the gate."""
    from oracle import Oracle
    sc = scenes.sand_column(8, 6, 8, with_block=False)
    sc.params = copy.copy(sc.params)
    sc.params.boundary_handling = abi.BOUNDARY_NONE
    sc.particles.position["point"][:, 1] -= np.float32(1.25 * sc.cell_width)     # the lowest two layers sit inside the floor cuboid (top at 3.5 h)
    dt, n = 2.5e-4, 80
    free = _hip(Pipeline, sc, n, dt)
    sc.collider_options = {"cuboid": [abi.collider_options(penalty_stiffness=2.0e5)]}
    with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.enable_penalty_force(True)
        p.substeps(sc.params, sc.gravity, dt, n)
        got = p.read_particles()
    refs = []
    for f64 in (False, True):
        o = Oracle(sc.cell_width, f64=f64, domain_log2_bytes=sc.domain_log2_bytes, threads=_threads(), penalty_force=True)
        o.load_scene(sc)
        for _ in range(n):
            o.substep_fixed(sc.params, sc.gravity, dt)
        refs.append(o.get_particles())
        o.close()
    rep, bad = parity_verdict(got, refs[0], refs[1], sc.cell_width)
    dy = float((got.position["point"][:, 1] - free.position["point"][:, 1]).mean()) / sc.cell_width
    print("penalty force: vs f64", rep["vs_f64"]["eps_x"], rep["vs_f64"]["eps_v"], "| mean lift over the free fall", dy, "h")
    assert not bad, (bad, rep)
    assert dy > 0.02, dy


def _two_floors(near, far):
    """Falling sand over TWO overlapping floor cuboids: `far`'s top lies 0.3 h below `near`'s, so for every node above the ground the projection on
    `near` is the closer one -- under the reference's GPU rule (src_kernels/cuda/grid_update.rs:110-146) only `near`'s handling may act."""
    sc = _falling_sand()
    lo = scenes.floor_cuboid(top_y=4 * sc.cell_width - 0.5 * sc.cell_width - 0.3 * sc.cell_width, friction=0.5)
    sc.colliders = [sc.colliders[0], lo]
    sc.collider_options = {"cuboid": [abi.collider_options(grid_boundary_handling=near), abi.collider_options(grid_boundary_handling=far)]}
    return sc


@pytest.mark.parametrize("near,far", [(abi.BOUNDARY_FRICTION, abi.BOUNDARY_STICK), (abi.BOUNDARY_NONE, abi.BOUNDARY_STICK), (abi.BOUNDARY_STICK, abi.BOUNDARY_FRICTION)])
def test_mixed_options_only_the_closest_collider_acts(Pipeline, near, far):
    """with per-collider options the reference picks the SINGLE closest projection over all colliders whose rule is not None and applies
    that collider's rule only (grid_update.rs:110-190); rounds 1 - 5 let every collider act in turn.  Two stacked floors with different rules:
    HIP vs oracle (which restates the GPU rule) under the gate, and the result must be the nearer floor's rule, not a mix."""
    sc = _two_floors(near, far)
    dt, n = 2.5e-4, 100
    got = _hip(Pipeline, sc, n, dt)
    r32, r64 = _oracles(sc, n, dt)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    assert not bad, (bad, rep)
    print("mixed options", near, far, "vs f64", rep["vs_f64"]["eps_x"], rep["vs_f64"]["eps_v"])
    if near == abi.BOUNDARY_NONE:
        # a collider whose rule is None is not a candidate at all (grid_update.rs:112-114): the far floor alone gives the same bits
        one = _falling_sand()
        one.colliders = [sc.colliders[1]]
        one.collider_options = {"cuboid": [abi.collider_options(grid_boundary_handling=far)]}
        alone = _hip(Pipeline, one, n, dt)
        assert np.abs(got.position["point"] - alone.position["point"]).max() / sc.cell_width < 1e-6


def test_gpu_rule_normal_is_the_sdf_gradient_on_a_heightfield(Pipeline):
    """With an option on the heightfield the collision normal is the central-difference gradient of the signed distance (grid_update.rs:72-95), which
    on a sinusoidal ground differs from the direction of the projection the CPU rule uses: HIP vs the oracle's GPU rule, and the two rules must
    really end apart."""
    sc = scenes.sand_on_heightfield(16, 6, 12, below_is_solid=True)
    sc.particles.velocity["vector"][:, 1] = -3.0
    sc.particles.velocity["vector"][:, 0] = 2.0
    dt, n = 2.5e-4, 120
    cpu_rule = _hip(Pipeline, sc, n, dt)
    sc.collider_options = {"heightfield": [abi.collider_options(grid_boundary_handling=abi.BOUNDARY_FRICTION)]}
    got = _hip(Pipeline, sc, n, dt)
    r32, r64 = _oracles(sc, n, dt)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    apart = np.abs(got.position["point"] - cpu_rule.position["point"]).max() / sc.cell_width
    print("gpu rule on a heightfield: vs f64", rep["vs_f64"]["eps_x"], rep["vs_f64"]["eps_v"], "| differs from the CPU rule by", apart, "h")
    assert not bad, (bad, rep)
    assert apart > 1e-4, apart


def test_options_are_validated_and_reset(Pipeline):
    sc = _falling_sand()
    with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        with pytest.raises(SparklHipError):
            p.set_collider_options("cuboid", [])                                   # one record per collider
        with pytest.raises(SparklHipError):
            p.set_collider_options("cuboid", [abi.collider_options(grid_boundary_handling=7)])
        with pytest.raises(SparklHipError):
            p.set_collider_options("heightfield", [abi.collider_options()])       # there is no heightfield
        p.set_collider_options("cuboid", [abi.collider_options(grid_boundary_handling=abi.BOUNDARY_NONE)])
        p.set_colliders(sc.colliders)                                              # set_* resets the options of its kind
        p.substeps(sc.params, sc.gravity, 2.5e-4, 100)
        got = p.read_particles()
    base = _hip(Pipeline, sc, 100, 2.5e-4)
    np.testing.assert_array_equal(got.position["point"], base.position["point"])


@pytest.mark.parametrize("kind", ["heightfield", "trimesh", "heightfield+gpu_rule", "trimesh+gpu_rule"])
def test_projection_cache_is_bit_identical_and_survives_rebins(Pipeline, kind, monkeypatch):
    """The cache replays the floats the triangle search returned: with it on (default) and off (SPARKL_HIP_PROJ_CACHE=0) the particle
    state after 240 substeps -- several re-bins, which rebuild the node blocks and clear the cache -- must be the same bits.
    "+gpu_rule" (round 6): the collider carries its own handling, so the grid update follows the reference's GPU rule and caches what the reference
    caches in the node (closest projection + signed-distance normal, src_kernels/cuda/grid_update.rs:110-146): seven collider scans per node once per
    binning instead of every substep."""
    gpu_rule = kind.endswith("+gpu_rule")
    kind = kind.split("+")[0]
    sc = scenes.sand_on_heightfield(16, 8, 12, below_is_solid=True) if kind == "heightfield" else scenes.sand_on_trimesh(16, 8, 12, oriented=True)
    if gpu_rule:
        sc.collider_options = {kind: [abi.collider_options(grid_boundary_handling=abi.BOUNDARY_FRICTION)]}
    sc.particles.velocity["vector"][:, 1] = -3.0
    sc.particles.velocity["vector"][:, 0] = 8.0
    dt, n = 2.5e-4, 240
    with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.load_scene(sc)
        p.clock_enable(True)
        p.substeps(sc.params, sc.gravity, dt, n)
        on = p.read_particles()
        rebins = int(p.clock_read().resorts)
    monkeypatch.setenv("SPARKL_HIP_PROJ_CACHE", "0")
    off = _hip(Pipeline, sc, n, dt)
    assert rebins >= 2
    for k in ("position", "velocity", "volume", "velocity_gradient"):
        assert getattr(on, k).tobytes() == getattr(off, k).tobytes(), k


def _sand_over_polyline():
    """A sand block over a V-shaped ground given as a closed counter-clockwise polygon in the x-y plane (extruded along z): the polyline collider of
    the reference's GPU path (gpu_collider.rs:99-201), rotated a little about z so that the pose is exercised."""
    sc = scenes.sand_column(12, 6, 10, with_block=False)
    sc.colliders = []
    sc.particles.velocity["vector"][:, 1] = -2.0
    sc.particles.velocity["vector"][:, 0] = 0.8
    verts = np.array([[-5.0, -1.0], [8.0, -1.0], [8.0, 1.2], [1.4, 0.75], [-5.0, 1.2]], dtype=np.float32)
    ang = 0.02
    sc.polylines = [abi.polyline(verts, translation=(0.0, 0.0, 0.3), rotation=(0.0, 0.0, np.sin(ang / 2), np.cos(ang / 2)), friction=0.4)]
    return sc


def test_polyline_collider_parity(Pipeline):
    sc = _sand_over_polyline()
    dt, n = 2.5e-4, 160
    got = _hip(Pipeline, sc, n, dt)
    r32, r64 = _oracles(sc, n, dt)
    rep, bad = parity_verdict(got, r32, r64, sc.cell_width)
    free_vy = -2.0 + sc.gravity[1] * dt * n
    touched = int((np.abs(r64.velocity["vector"][:, 1] - free_vy) > 0.05).sum())
    print("polyline ground: vs f64", rep["vs_f64"]["eps_x"], rep["vs_f64"]["eps_v"], "| particles the ground has slowed:", touched)
    assert touched > 100, "the scene must reach the collider"
    assert not bad, (bad, rep)


def test_polyline_inside_test_matches_the_oracle_node_by_node(Pipeline):
    """Stick handling makes is_inside visible on the grid: node velocities after one substep against the oracle's, including the nodes next to
    the reentrant vertex of the V (where the inside test takes the adjacent segment's direction)."""
    from oracle import Oracle
    from tests.parity import node_errors
    sc = _sand_over_polyline()
    sc.params = copy.copy(sc.params)
    sc.params.boundary_handling = abi.BOUNDARY_STICK
    sc.particles.position["point"][:, 1] -= np.float32(0.4 * sc.cell_width)     # the lowest particles scatter onto nodes inside the ground
    dt = 2.5e-4
    o = Oracle(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, threads=_threads())
    o.load_scene(sc)
    o.sort(); o.p2g(dt, sc.gravity); o.grid_update(dt, sc.params.boundary_handling, 0)
    ref = o.nodes()
    with Pipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
        p.grid_debug_capture(True)
        p.load_scene(sc)
        p.substeps(sc.params, sc.gravity, dt, 1)
        e = node_errors(p.read_grid_nodes(), ref)
    stuck = int((np.abs(ref["velocity"]).max(axis=1) == 0.0).sum())
    print("polyline stick: node errors", e, "| nodes stopped inside the ground:", stuck)
    assert e["n_got"] == e["n_ref"] == e["n_common"]
    assert stuck > 20
    assert e["velocity_rel"] < 2e-5
