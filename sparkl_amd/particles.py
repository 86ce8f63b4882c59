"""`ParticleSet` / `Particle::new` / `extract_particles_components` host mirrors (SoA on the host too).

reference: src/dynamics/particle.rs:54-103 (Particle::new defaults), src/dynamics/particle_set.rs:11-73,
           src/cuda/cuda_particle_set.rs:179-248 (ParticleComponents / extract_particles_components).
"""
import numpy as np

from . import abi

f32 = np.float32


class ParticleComponents:
    """`ParticleComponents` (cuda_particle_set.rs:179-186) + the CPU path's velocity_gradient."""

    __slots__ = ("status", "position", "velocity", "volume", "phase", "data", "velocity_gradient")

    def __init__(self, n):
        self.status = np.zeros(n, abi.STATUS_DTYPE)
        self.position = np.zeros(n, abi.POSITION_DTYPE)
        self.velocity = np.zeros(n, abi.VELOCITY_DTYPE)
        self.volume = np.zeros(n, abi.VOLUME_DTYPE)
        self.phase = np.zeros(n, abi.PHASE_DTYPE)
        self.data = np.zeros(n, abi.DATA_DTYPE)
        self.velocity_gradient = np.zeros(n, abi.VELGRAD_DTYPE)

    def __len__(self):
        return len(self.status)

    def copy(self):
        c = ParticleComponents(0)
        for k in self.__slots__:
            setattr(c, k, getattr(self, k).copy())
        return c

    @staticmethod
    def concatenate(parts):
        c = ParticleComponents(0)
        for k in ParticleComponents.__slots__:
            setattr(c, k, np.concatenate([getattr(p, k) for p in parts]))
        return c


def new_particles(model, positions, radius, density0):
    """Vectorised `Particle::new(model, position, radius, density0)` (src/dynamics/particle.rs:54-103)."""
    positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
    n = len(positions)
    c = ParticleComponents(n)
    c.position["point"] = positions
    volume0 = (f32(radius) * f32(2.0)) ** 3  # (radius * 2.0).powi(DIM)
    volume0 = f32(f32(radius) * f32(2.0)) * f32(f32(radius) * f32(2.0)) * f32(f32(radius) * f32(2.0))
    c.volume["volume0"] = volume0
    c.volume["mass"] = f32(volume0 * f32(density0))
    c.volume["radius0"] = f32(radius)
    c.volume["deformation_gradient"] = np.eye(3, dtype=f32).reshape(9)
    c.volume["plastic_deformation_gradient_det"] = 1.0
    c.status["model_index"] = model
    c.phase["phase"] = 1.0
    c.data["nacc_alpha"] = -0.01
    c.data["plastic_hardening"] = 1.0
    c.data["elastic_hardening"] = 1.0
    return c


class ParticleSet:
    """`ParticleSet::{new, len, insert_batch, remove_range}` (src/dynamics/particle_set.rs:20-73)."""

    def __init__(self):
        self.components = ParticleComponents(0)

    def __len__(self):
        return len(self.components)

    def insert_batch(self, comps):
        self.components = ParticleComponents.concatenate([self.components, comps])

    def remove_range(self, start, end):
        keep = np.r_[0:start, end:len(self)]
        c = ParticleComponents(0)
        for k in ParticleComponents.__slots__:
            setattr(c, k, getattr(self.components, k)[keep].copy())
        self.components = c


def extract_particles_components(particles):
    """`extract_particles_components(&[Particle])` (cuda_particle_set.rs:189-248)."""
    return particles.components
