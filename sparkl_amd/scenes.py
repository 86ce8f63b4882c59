"""Synthetic inputs of SURVEY.md section 8(d): the common lattice generator and the five BASELINE configs.

`cube_particles` mirrors examples3d/helper.rs:6-40 (lattice at spacing 2*radius, i-major / k-minor order);
the jitter is PCG32(seed=42, stream=54) drawn in lattice order, +-0.25 spacing per axis.
"""
import numpy as np

from . import abi
from .models import (CorotatedLinearElasticity, DruckerPragerPlasticity, MonaghanSphEos,
                     NeoHookeanElasticity, ParticleModel, ParticleModelSet)
from .particles import ParticleComponents, new_particles

f32 = np.float32
_MULT = np.uint64(6364136223846793005)


def pcg32_stream(n, seed=42, stream=54):
    """First n outputs (uint32) of PCG32 XSH-RR with the reference seeding sequence, vectorised by LCG jumps."""
    mask = (1 << 64) - 1
    mult = int(_MULT)
    inc = ((stream << 1) | 1) & mask
    state = 0
    state = (state * mult + inc) & mask
    state = (state + seed) & mask
    state = (state * mult + inc) & mask
    blk = 1 << 16
    # (A_k, C_k): state_{k} = A_k * s0 + C_k
    A = np.empty(blk, np.uint64)
    Cc = np.empty(blk, np.uint64)
    a, c = 1, 0
    for k in range(blk):
        A[k], Cc[k] = a, c
        a, c = (a * mult) & mask, (c * mult + inc) & mask
    jump_a, jump_c = a, c
    out = np.empty(n, np.uint32)
    with np.errstate(over="ignore"):
        for start in range(0, n, blk):
            m = min(blk, n - start)
            old = A[:m] * np.uint64(state) + Cc[:m]
            xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
            rot = (old >> np.uint64(59)).astype(np.uint32)
            out[start:start + m] = (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))
            state = (state * jump_a + jump_c) & mask
    return out


def cube_positions(origin, ni, nj, nk, radius, jitter=True, rng_offset=0, seed=42):
    """Lattice positions (i-major) at spacing s = 2*radius, optional +-0.25 s PCG32 jitter."""
    s = f32(radius) * f32(2.0)
    i, j, k = np.meshgrid(np.arange(ni, dtype=f32), np.arange(nj, dtype=f32), np.arange(nk, dtype=f32), indexing="ij")
    pos = np.stack([i.ravel() * s, j.ravel() * s, k.ravel() * s], axis=1).astype(f32)
    pos += np.asarray(origin, dtype=f32)[None, :]
    if jitter:
        n = ni * nj * nk
        r = pcg32_stream(3 * (n + rng_offset), seed=seed)[3 * rng_offset:]
        u = (r >> np.uint32(8)).astype(f32) * f32(2.0 ** -24)          # [0, 1)
        pos += ((u - f32(0.5)) * f32(0.5) * s).reshape(n, 3).astype(f32)
    return np.ascontiguousarray(pos, dtype=f32)


def floor_cuboid(top_y, friction=0.5, half=1000.0, thickness=50.0):
    c = abi.CuboidColliderC()
    c.half_extents[:] = [half, thickness, half]
    c.rotation[:] = [0.0, 0.0, 0.0, 1.0]
    c.translation[:] = [0.0, top_y - thickness, 0.0]
    c.friction = friction
    return c


class Scene:
    def __init__(self, name, cell_width, models, particles, colliders, params, gravity, fixed_dt, domain_log2_bytes=30):
        self.name, self.cell_width, self.models, self.particles = name, cell_width, models, particles
        self.colliders, self.params, self.gravity, self.fixed_dt = colliders, params, gravity, fixed_dt
        self.domain_log2_bytes = domain_log2_bytes
        self.heightfields = []
        self.trimeshes = []
        self.polylines = []
        self.collider_options = {}     # kind ("cuboid" | "heightfield" | "trimesh") -> [abi.collider_options(..)] (CudaColliderOptions)


def _domain_bits_for(max_cell):
    for bits, dom in ((30, 256), (33, 512), (36, 1024), (39, 2048)):
        if max_cell + 8 < dom:
            return bits
    raise ValueError("scene does not fit the largest virtual grid")


def sand_column(nx, ny, nz, h=0.2, with_block=False, jitter=True, name=None, x_slab=(0, 1), stock_lattice=False):
    """Configs 1/3: Drucker-Prager sand (corotated, E=1e7, nu=0.2, rho=2700) on a friction floor.

    with_block=True stacks the sand on an elastic block of the same size (config 1, examples3d/sand3.rs:39-78).
    x_slab=(k, K): generate only the k-th of K x-slabs of a column K*nx particles long (multi-GPU weak scaling).
    stock_lattice=True: the reference examples' own sampling (examples3d/helper.rs:28-33 with randomize = false, sand3.rs:49-71): no jitter and
    the lattice origin on a multiple of the cell width, so that every other particle of a row sits ON a half-cell tie of
    `round(x / h)` (sp_grid.rs:216-226) -- the scenes above sample a quarter cell off, where no tie can occur.
    """
    E, NU = 1.0e7, 0.2
    models = ParticleModelSet()
    sand = models.insert(ParticleModel.with_plasticity(CorotatedLinearElasticity(E, NU), DruckerPragerPlasticity(E, NU)))
    r = h / 4.0
    origin = np.array([4 * h, 4 * h, 4 * h], dtype=f32) + (f32(0.0) if stock_lattice else f32(r))  # >= 2 cells inside the domain
    if stock_lattice:
        jitter = False
    origin[0] += f32(x_slab[0] * nx) * f32(2 * r)
    parts = []
    y0 = origin[1]
    if with_block:
        block = models.insert(ParticleModel.new(CorotatedLinearElasticity(E, NU)))
        parts.append(new_particles(block, cube_positions(origin, nx, ny, nz, r, jitter), r, 2700.0))
        y0 = origin[1] + f32(ny) * f32(2 * r)
        parts.append(new_particles(sand, cube_positions([origin[0], y0, origin[2]], nx, ny, nz, r, jitter, rng_offset=nx * ny * nz), r, 2700.0))
    else:
        parts.append(new_particles(sand, cube_positions(origin, nx, ny, nz, r, jitter, seed=42 + x_slab[0]), r, 2700.0))
    comps = ParticleComponents.concatenate(parts)
    floor = floor_cuboid(top_y=4 * h - 0.5 * h + 0.0, friction=0.5)
    params = abi.SolverParameters()
    max_cell = int(np.ceil(max(comps.position["point"].max(), (4 * h + x_slab[1] * nx * 2 * r)) / h)) + 2
    return Scene(name or f"sand_column_{nx}x{ny}x{nz}", h, models, comps, [floor], params, (0.0, -9.81, 0.0), 2.5e-4,
                 _domain_bits_for(max_cell))


def elastic_cube(n, h=0.2, jitter=True, name=None, stock_lattice=False):
    """Config 2: neo-Hookean cube n^3 (E=1e6, nu=0.3, rho=1000), no gravity/collider, spinning + expanding."""
    models = ParticleModelSet()
    mid = models.insert(ParticleModel.new(NeoHookeanElasticity(1.0e6, 0.3)))
    r = h / 4.0
    origin = np.array([4 * h, 4 * h, 4 * h], dtype=f32) + (f32(0.0) if stock_lattice else f32(r))
    pos = cube_positions(origin, n, n, n, r, jitter and not stock_lattice)
    comps = new_particles(mid, pos, r, 1000.0)
    c = pos.mean(axis=0, dtype=np.float64).astype(f32)
    d = pos - c
    omega = np.array([0.0, 0.0, 2.0], dtype=f32)
    comps.velocity["vector"] = (np.cross(np.broadcast_to(omega, d.shape), d) + f32(0.1) * d).astype(f32)
    params = abi.SolverParameters()
    params.boundary_handling = abi.BOUNDARY_NONE
    max_cell = int(np.ceil(pos.max() / h)) + 2
    return Scene(name or f"elastic_cube_{n}", h, models, comps, [], params, (0.0, 0.0, 0.0), 1.0e-4, _domain_bits_for(max_cell))


def block_on_floor(n, constitutive, plasticity=None, failure=None, density=1000.0, v0=(0.0, -3.0, 0.0), expand=0.0, h=0.2, dt=1.0e-4,
                   gravity=(0.0, -9.81, 0.0), name="block", jitter=True):
    """An n^3 block one and a half cells above a friction floor, falling at v0 and/or expanding at `expand` 1/s: the small scenes
    that exercise the plastic / failure models the 3-D examples do not use."""
    models = ParticleModelSet()
    mid = models.insert(ParticleModel(constitutive, plastic_model=plasticity, failure_model=failure))
    r = h / 4.0
    origin = np.array([4 * h, 5 * h, 4 * h], dtype=f32) + f32(r)
    pos = cube_positions(origin, n, n, n, r, jitter)
    comps = new_particles(mid, pos, r, density)
    c = pos.mean(axis=0, dtype=np.float64).astype(f32)
    d = pos - c
    half = f32(0.5 * n * 2 * r)
    # expansion rate grows linearly from the centre to `expand` at the faces: the strain, hence the yield / failure front, varies in space
    comps.velocity["vector"] = (np.asarray(v0, dtype=f32)[None, :] + f32(expand) * d * (np.abs(d) / half)).astype(f32)
    floor = floor_cuboid(top_y=4 * h - 0.5 * h, friction=0.5)
    params = abi.SolverParameters()
    max_cell = int(np.ceil(pos.max() / h)) + 6
    return Scene(f"{name}_{n}", h, models, comps, [floor], params, gravity, dt, _domain_bits_for(max_cell))


def fluid_dam(nx, ny, nz, h=0.8, r=0.1, jitter=True, name=None, stock_lattice=False):
    """Config 4: fluids3-like dam break (examples3d/fluids3.rs:17-43), 64 particles per cell.
    stock_lattice=True: fluids3's own sampling -- origin on a multiple of the cell width (it uses 2 h), no jitter: every fourth particle of a row
    sits on a half-cell tie (see sand_column)."""
    models = ParticleModelSet()
    mid = models.insert(ParticleModel.new(MonaghanSphEos(1.0e6, 7, 1.01e-3)))
    origin = np.array([4 * h, 4 * h, 4 * h], dtype=f32) + (f32(0.0) if stock_lattice else f32(r))
    pos = cube_positions(origin, nx, ny, nz, r, jitter and not stock_lattice)
    comps = new_particles(mid, pos, r, 1000.0)
    floor = floor_cuboid(top_y=4 * h - 0.5 * h, friction=0.0)
    params = abi.SolverParameters()
    params.force_fluids_volume_recomputation = 1
    max_cell = int(np.ceil(pos.max() / h)) + 2
    return Scene(name or f"fluid_dam_{nx}x{ny}x{nz}", h, models, comps, [floor], params, (0.0, -9.81, 0.0), 1.0e-4,
                 _domain_bits_for(max_cell))


def cube_through_sand(nx, ny, nz, cube_n=25, h=0.2, jitter=True, name=None, x_slab=(0, 1)):
    """Config 5: sand bed + kinematic cube moving at (10,0,0) (examples3d/cube_through_sand3.rs:40-81).

    x_slab=(k, K): only the k-th of K x-slabs of a bed K*nx particles long (multi-GPU); the cube starts over slab 0 and belongs
    to it.  `Scene.id_offset` is then the global id of the slab's first particle (bed slabs in order, the cube after slab 0)."""
    k, K = x_slab
    sc = sand_column(nx, ny, nz, h=h, jitter=jitter, x_slab=x_slab)
    r = h / 4.0
    kin_model = sc.models.insert(ParticleModel.new(CorotatedLinearElasticity(1.0e7, 0.2)))
    n_bed = nx * ny * nz
    nc = cube_n ** 3
    if K == 1:
        top = sc.particles.position["point"][:, 1].max()
    else:   # the same on every rank: the lattice top plus the largest jitter
        top = f32(4 * h + r) + f32(ny - 1) * f32(2 * r) + f32(0.25 * 2 * r)
    if k == 0:
        origin = [4 * h + r, top + 2 * h, 4 * h + r + (nz // 2 - cube_n // 2) * 2 * r]
        kin = new_particles(kin_model, cube_positions(origin, cube_n, cube_n, cube_n, r, jitter, rng_offset=nx * ny * nz), r, 2700.0)
        kin.status["kinematic_vel_enabled"] = 1
        kin.status["kinematic_vel"] = np.array([10.0, 0.0, 0.0], dtype=f32)
        sc.particles = ParticleComponents.concatenate([sc.particles, kin])
    sc.id_offset = k * n_bed + (nc if k > 0 else 0)
    sc.total_particles = K * n_bed + nc
    sc.name = name or f"cube_through_sand_{K * nx}x{ny}x{nz}"
    if K == 1:
        max_cell = int(np.ceil(sc.particles.position["point"].max() / h)) + 2
    else:
        max_cell = int(np.ceil(max(float(sc.particles.position["point"].max()), float(top) + 2 * h + cube_n * 2 * r, 4 * h + K * nx * 2 * r) / h)) + 2
    sc.domain_log2_bytes = _domain_bits_for(max_cell)
    return sc


def sand_on_heightfield(nx, ny, nz, h=0.2, jitter=True, below_is_solid=False):
    """sand3-like scene with its real ground (examples3d/sand3.rs:23-37): a 41 x 41 sinusoidal heightfield instead of the
    cuboid floor.  Scaled down so that the particle cloud sits over the middle of one valley."""
    sc = sand_column(nx, ny, nz, h=h, jitter=jitter, name=f"sand_on_heightfield_{nx}x{ny}x{nz}")
    n = 40
    i = np.arange(n + 1, dtype=f32)
    heights = np.repeat((-np.sin(i * f32(np.pi) / f32(n)))[:, None], n + 1, axis=1).astype(f32)   # heights[(i, j)] = -sin(i pi / n)
    side = 16.0
    lo = sc.particles.position["point"].min(0)
    cx = float(sc.particles.position["point"][:, 0].mean())
    cz = float(sc.particles.position["point"][:, 2].mean())
    # surface y = ty + scale_y * height ; put the valley bottom (height -1) two cells under the cloud
    ty = float(lo[1]) - 2.0 * h + 1.5
    sc.colliders = []
    sc.heightfields = [abi.heightfield(heights, (side, 1.5, side), translation=(cx, ty, cz), friction=0.5, below_is_solid=below_is_solid)]
    return sc


def box_mesh(half_extents):
    """A closed, outward-oriented triangle mesh of the box [-he, he]: 8 vertices, 12 triangles."""
    hx, hy, hz = [float(x) for x in half_extents]
    v = np.array([[-hx, -hy, -hz], [hx, -hy, -hz], [hx, hy, -hz], [-hx, hy, -hz], [-hx, -hy, hz], [hx, -hy, hz], [hx, hy, hz], [-hx, hy, hz]], dtype=f32)
    i = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [3, 7, 6], [3, 6, 2], [0, 4, 7], [0, 7, 3], [1, 2, 6], [1, 6, 5]], dtype=np.uint32)
    return v, i


def sand_on_trimesh(nx, ny, nz, h=0.2, jitter=True, oriented=True):
    """The sand column over a tilted wedge given as a triangle mesh (a rotated box): the third collider shape of
    cuda_collider_set.rs:88-102."""
    sc = sand_column(nx, ny, nz, h=h, jitter=jitter, name=f"sand_on_trimesh_{nx}x{ny}x{nz}")
    lo, hi = sc.particles.position["point"].min(0), sc.particles.position["point"].max(0)
    v, i = box_mesh([3.0, 0.5, 3.0])
    ang = 0.35                                        # rotation about z: the sand slides down the slope
    q = (0.0, 0.0, float(np.sin(ang / 2)), float(np.cos(ang / 2)))
    centre = (float(0.5 * (lo[0] + hi[0])), float(lo[1]) - 1.5 * h - 0.5 / np.cos(ang), float(0.5 * (lo[2] + hi[2])))
    sc.colliders = []
    sc.trimeshes = [abi.trimesh(v, i, translation=centre, rotation=q, friction=0.5, oriented=oriented)]
    return sc


# The BASELINE.json configs, by index
def baseline_config(i, scale=1.0):
    if i == 0:
        return sand_column(50, 25, 20, with_block=True, name="config1_sand3_50k")
    if i == 1:
        n = max(4, int(round(100 * scale ** (1 / 3))))
        return elastic_cube(n, name=f"config2_neo_hookean_{n}^3")
    if i == 2:
        f = scale ** (1 / 3)
        return sand_column(max(4, int(200 * f)), max(4, int(250 * f)), max(4, int(200 * f)), name="config3_drucker_prager_column")
    if i == 3:
        f = scale ** (1 / 3)
        return fluid_dam(max(8, int(500 * f)), max(8, int(200 * f)), max(8, int(200 * f)), name="config4_fluid_dam")
    if i == 4:
        f = scale ** (1 / 3)
        return cube_through_sand(max(8, int(800 * f)), max(8, int(250 * f)), max(8, int(400 * f)), name="config5_cube_through_sand")
    raise IndexError(i)
