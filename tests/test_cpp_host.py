"""The C++ host mirror (include/sparkl_hip.hpp): compiles against the C ABI here; on the GPU box it is run and compared
with the Python host mirror driving the same library on the same scene."""
import os
import subprocess

import numpy as np
import pytest

from sparkl_amd import _lib, abi, scenes
from sparkl_amd.models import CorotatedLinearElasticity, DruckerPragerPlasticity, ParticleModel, ParticleModelSet
from sparkl_amd.particles import ParticleComponents, new_particles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_example")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "host_example.cpp")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", EXE, "-L", libdir, "-lsparkl_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_host_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_matches_python_host():
    _build()
    nsub = 10
    out = subprocess.check_output([EXE, str(nsub)], text=True)
    print(out)
    line = [l for l in out.splitlines() if l.startswith("n ")][0].split()
    vals = {line[i]: line[i + 1] for i in (0, 2, 4)}
    mean_pos = np.array([float(x) for x in line[7:10]])
    mean_vy, mean_fyy = float(line[11]), float(line[13])
    assert "custom: rejected code 2" in out
    assert "slabs:" in out and out.split("slabs:")[1].splitlines()[0].rstrip().endswith("ok")   # two slabs stepped inside the library, no Python
    assert "readback: snapshot" in out and out.split("readback:")[1].splitlines()[0].rstrip().endswith("ok")
    assert "grid check: ok" in out    # sparkl_hip_grid_check (the reference's check_active_blocks) through the C++ mirror
    # same scene through the Python mirror
    h, r = 0.2, 0.05
    models = ParticleModelSet()
    sand = models.insert(ParticleModel.with_plasticity(CorotatedLinearElasticity(1.0e7, 0.2), DruckerPragerPlasticity(1.0e7, 0.2)))
    block = models.insert(ParticleModel.new(CorotatedLinearElasticity(1.0e7, 0.2)))
    x0 = np.float32(4 * 0.2) + np.float32(r)
    top = scenes.cube_positions([x0, x0 + np.float32(10 * r * 2), x0], 16, 10, 12, r, jitter=False)
    bot = scenes.cube_positions([x0, x0, x0], 16, 10, 12, r, jitter=False)
    comps = ParticleComponents.concatenate([new_particles(sand, top, r, 2700.0), new_particles(block, bot, r, 2700.0)])
    from sparkl_amd.pipeline import MpmPipeline
    params = abi.SolverParameters()
    with MpmPipeline(h) as p:
        p.set_models(models)
        p.set_colliders([scenes.floor_cuboid(top_y=4 * h - 0.5 * h, friction=0.5)])
        p.upload_particles(comps)
        p.substeps(params, (0, -9.81, 0), 2.5e-4, nsub)
        t, tim = p.step(params, (0, -9.81, 0), timings=True)
        got = p.read_particles()
    assert int(vals["n"]) == len(comps)
    assert float(vals["simulated"]) == pytest.approx(t, rel=1e-6) and int(vals["substeps"]) == tim.num_substeps
    assert np.allclose(mean_pos, got.position["point"].astype(np.float64).mean(0), rtol=1e-6)
    assert mean_vy == pytest.approx(got.velocity["vector"][:, 1].astype(np.float64).mean(), rel=2e-3)
    assert mean_fyy == pytest.approx(got.volume["deformation_gradient"][:, 4].astype(np.float64).mean(), rel=1e-6)
