"""The slab scene builders behind `bench.py --gpus N` (CPU only): the N > 1 job is ONE scene whatever N when the scaling is strong,
ids are contiguous across the slabs, the kinematic cube of configs[4] belongs to slab 0, and every slab is a whole number of
4-cell blocks wide."""
import argparse

import numpy as np
import pytest

import bench
from sparkl_amd import scenes


def test_strong_dims_do_not_depend_on_the_rank_count():
    for total, base in ((10.0e6, (200, 250, 200)), (80.0e6, (800, 250, 400)), (2.0e5, (800, 250, 400))):
        dims = {bench.strong_dims(total, w, base=base) for w in (1, 2, 4, 8)}
        assert len(dims) == 1, dims
        nx, ny, nz = dims.pop()
        assert nx % 64 == 0 and abs(nx * ny * nz - total) / total < 0.05
    assert bench.strong_dims(80.0e6, 8, base=(800, 250, 400)) == (768, 260, 400)
    assert bench.strong_dims(10.0e6, 8) == (192, 260, 200)


def test_cube_through_sand_slabs_are_one_scene():
    nx, ny, nz, cube_n, K = 64, 10, 12, 4, 4
    slabs = [scenes.cube_through_sand(nx // K, ny, nz, cube_n=cube_n, x_slab=(k, K)) for k in range(K)]
    h = slabs[0].cell_width
    assert all(s.total_particles == nx * ny * nz + cube_n ** 3 for s in slabs)
    assert sum(len(s.particles) for s in slabs) == slabs[0].total_particles
    # ids: slab 0 = its bed + the cube, the others follow
    off = 0
    for s in slabs:
        assert s.id_offset == off
        off += len(s.particles)
    kin = [int((s.particles.status["kinematic_vel_enabled"] != 0).sum()) for s in slabs]
    assert kin == [cube_n ** 3, 0, 0, 0]
    # x ranges of the bed slabs: disjoint, in order, each a whole number of blocks (8 particles = 4 cells) wide
    assert (nx // K) % 8 == 0
    for k, s in enumerate(slabs):
        bed = s.particles.position["point"][s.particles.status["kinematic_vel_enabled"] == 0]
        cells = np.round(bed[:, 0] / np.float32(h)).astype(int) - 1
        lo, hi = 4 + k * (nx // K) // 2, 4 + (k + 1) * (nx // K) // 2
        assert cells.min() >= lo - 1 and cells.max() <= hi, (k, cells.min(), cells.max(), lo, hi)
    assert len({s.domain_log2_bytes for s in slabs}) == 1, "every rank must build the same virtual grid"


def test_bench_arguments():
    a = bench.parse_args(["--config", "cube80m", "--gpus", "8"])
    assert a.scaling == "strong"
    a = bench.parse_args(["--gpus", "8"])
    # round 6: the default N > 1 job is the metric's own -- the 10 M column, fixed, cut N ways (BASELINE.json: ">= 6x strong scaling at 8 GPUs")
    assert a.scaling == "strong" and a.config == "sand10m" and a.weak_figure and a.strong_projection
    a = bench.parse_args(["--gpus", "8", "--scaling", "weak"])
    assert a.scaling == "weak"
    assert bench.parse_args(["--config", "fluid20m"]).scaling == "weak"
    with pytest.raises(SystemExit):
        bench.parse_args(["--config", "cube80m", "--scaling", "weak"])
    with pytest.raises(SystemExit):
        bench.parse_args(["--config", "neo1m", "--scaling", "strong"])
