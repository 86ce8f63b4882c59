"""The C ABI: struct sizes match the Rust repr(C) layouts quoted in SURVEY appendix A, the library loads and
exports every function the header declares (no compute calls here: no GPU in this container)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from sparkl_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sparkl_hip.h")


def test_python_mirror_sizes():
    # src_core/dynamics/particle.rs sizes: 12 / 12 / 52 / 24 / 8 / 16
    assert abi.POSITION_DTYPE.itemsize == 12 and abi.VELOCITY_DTYPE.itemsize == 12
    assert abi.VOLUME_DTYPE.itemsize == 52 and abi.STATUS_DTYPE.itemsize == 24
    assert abi.PHASE_DTYPE.itemsize == 8 and abi.DATA_DTYPE.itemsize == 16
    assert abi.STATUS_DTYPE.fields["kinematic_vel"][1] == 4 and abi.STATUS_DTYPE.fields["model_index"][1] == 16
    assert C.sizeof(abi.SolverParameters) == 28  # solver_parameters.rs:42-54
    assert C.sizeof(abi.ConstitutiveModelC) == 20 and C.sizeof(abi.PlasticModelC) == 36 and C.sizeof(abi.FailureModelC) == 12


def test_header_compiles_as_c_and_sizes_agree():
    src = r'''
#include <stdio.h>
#include "sparkl_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(SparklParticlePosition), sizeof(SparklParticleVelocity),
    sizeof(SparklParticleVolume), sizeof(SparklParticleStatus), sizeof(SparklParticlePhase), sizeof(SparklParticleData),
    sizeof(SparklVelocityGradient), sizeof(SparklSolverParams), sizeof(SparklConstitutiveModel), sizeof(SparklPlasticModel),
    sizeof(SparklFailureModel), sizeof(SparklParticleModel), sizeof(SparklCuboidCollider), sizeof(SparklGridNode));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [12, 12, 52, 24, 8, 16, 36, 28, 20, 36, 12, C.sizeof(abi.ParticleModelC), C.sizeof(abi.CuboidColliderC),
                     abi.GRID_NODE_DTYPE.itemsize]


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sparkl_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib):
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(hip_lib, n), f"libsparkl_hip.so does not export {n}"
    assert sorted(_lib.SYMBOLS) == names
    assert b"gfx950" in hip_lib.sparkl_hip_version()


def test_create_fails_loudly_without_gpu(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sparkl_amd.pipeline import MpmPipeline
    with pytest.raises(_lib.SparklHipError) as e:
        MpmPipeline(0.2)
    assert e.value.code == 3  # SPARKL_ERR_NO_DEVICE: no silent CPU fallback
