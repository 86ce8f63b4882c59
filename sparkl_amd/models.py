"""Host-side mirrors of sparkl's material models (constructors + `to_core_model()` bridge).

Same names, arguments and defaults as the reference:
  CorotatedLinearElasticity::new   src_core/dynamics/models/elasticity_corotated_linear.rs:19-29
  NeoHookeanElasticity::new        src_core/dynamics/models/elasticity_neo_hookean.rs:17-26
  MonaghanSphEos::new              src_core/dynamics/models/eos_monaghan_sph.rs:19-26
  DruckerPragerPlasticity::new     src_core/dynamics/models/plasticity_drucker_prager.rs:22-34
  ParticleModel / ParticleModelSet src/dynamics/particle_model.rs:14-18,85-139
Lame parameters are computed in f32 exactly as src_core/utils/physics.rs:18-27.
"""
import numpy as np

from . import abi

f32 = np.float32


def lame_lambda_mu(young_modulus, poisson_ratio):
    """physics.rs:18-27 in f32."""
    e, nu = f32(young_modulus), f32(poisson_ratio)
    lam = e * nu / ((f32(1.0) + nu) * (f32(1.0) - f32(2.0) * nu))
    mu = e / (f32(2.0) * (f32(1.0) + nu))
    return float(lam), float(mu)


class CorotatedLinearElasticity:
    def __init__(self, young_modulus, poisson_ratio):
        self.lambda_, self.mu = lame_lambda_mu(young_modulus, poisson_ratio)
        self.split_stress_on_failure = True
        self.cfl_coeff = 0.9

    def is_fluid(self):
        return False

    def to_core_model(self):
        m = abi.ConstitutiveModelC()
        m.tag = abi.CONSTITUTIVE_COROTATED_LINEAR
        m.u.corotated.split_stress_on_failure = int(self.split_stress_on_failure)
        m.u.corotated.cfl_coeff = self.cfl_coeff
        m.u.corotated.lambda_ = self.lambda_
        m.u.corotated.mu = self.mu
        return m


class NeoHookeanElasticity:
    def __init__(self, young_modulus, poisson_ratio):
        self.lambda_, self.mu = lame_lambda_mu(young_modulus, poisson_ratio)
        self.cfl_coeff = 0.5

    def is_fluid(self):
        return False

    def to_core_model(self):
        m = abi.ConstitutiveModelC()
        m.tag = abi.CONSTITUTIVE_NEO_HOOKEAN
        m.u.neo_hookean.cfl_coeff = self.cfl_coeff
        m.u.neo_hookean.lambda_ = self.lambda_
        m.u.neo_hookean.mu = self.mu
        return m


class MonaghanSphEos:
    def __init__(self, pressure0, gamma, viscosity):
        self.pressure0, self.gamma, self.viscosity = pressure0, int(gamma), viscosity
        self.max_neg_pressure = 1.0

    def is_fluid(self):
        return True

    def to_core_model(self):
        m = abi.ConstitutiveModelC()
        m.tag = abi.CONSTITUTIVE_EOS_MONAGHAN_SPH
        m.u.monaghan.pressure0 = self.pressure0
        m.u.monaghan.gamma = self.gamma
        m.u.monaghan.viscosity = self.viscosity
        m.u.monaghan.max_neg_pressure = self.max_neg_pressure
        return m


class DruckerPragerPlasticity:
    def __init__(self, young_modulus, poisson_ratio):
        self.lambda_, self.mu = lame_lambda_mu(young_modulus, poisson_ratio)
        # (35.0 as Real).to_radians() etc. in f32
        self.h0 = float(np.deg2rad(f32(35.0), dtype=f32))
        self.h1 = float(np.deg2rad(f32(9.0), dtype=f32))
        self.h2 = 0.2
        self.h3 = float(np.deg2rad(f32(10.0), dtype=f32))
        self.only_active_when_failed = False
        self.volume_correction = 1.0

    def to_core_model(self):
        m = abi.PlasticModelC()
        m.tag = abi.PLASTIC_DRUCKER_PRAGER
        d = m.u.drucker_prager
        d.h0, d.h1, d.h2, d.h3 = self.h0, self.h1, self.h2, self.h3
        d.lambda_, d.mu = self.lambda_, self.mu
        d.only_active_when_failed = int(self.only_active_when_failed)
        d.volume_correction = self.volume_correction
        return m


class SnowPlasticity:
    """`SnowPlasticity::new` (plasticity_snow.rs:18-24): min_epsilon 2.5e-2, max_epsilon 4.5e-3, hardening_coeff 10."""

    def __init__(self):
        self.min_epsilon, self.max_epsilon, self.hardening_coeff = 2.5e-2, 4.5e-3, 10.0

    def to_core_model(self):
        m = abi.PlasticModelC()
        m.tag = abi.PLASTIC_SNOW
        m.u.snow.min_epsilon, m.u.snow.max_epsilon, m.u.snow.hardening_coeff = self.min_epsilon, self.max_epsilon, self.hardening_coeff
        return m


class RankinePlasticity:
    """`RankinePlasticity::new(young_modulus, poisson_ratio, tensile_strength, softening_rate)` (plasticity_rankine.rs:19-32)."""

    def __init__(self, young_modulus, poisson_ratio, tensile_strength, softening_rate):
        self.lambda_, self.mu = lame_lambda_mu(young_modulus, poisson_ratio)
        self.tensile_strength, self.softening_rate = tensile_strength, softening_rate

    def to_core_model(self):
        m = abi.PlasticModelC()
        m.tag = abi.PLASTIC_RANKINE
        r = m.u.rankine
        r.mu, r.lambda_, r.tensile_strength, r.softening_rate = self.mu, self.lambda_, self.tensile_strength, self.softening_rate
        return m


class NaccPlasticity:
    """`NaccPlasticity::new(young_modulus, poisson_ratio, cohesion, hardening_enabled, hardening_factor, friction_angle)`
    (plasticity_nacc.rs:22-43); `with_m` (:45-60) through the `m=` keyword.  dim3: d = 3."""

    def __init__(self, young_modulus, poisson_ratio, cohesion, hardening_enabled, hardening_factor, friction_angle=None, m=None):
        e, nu = f32(young_modulus), f32(poisson_ratio)
        self.mu = float(e / (f32(2.0) * (f32(1.0) + nu)))               # physics.rs:25-27
        self.kappa = float(e / (f32(3.0) * (f32(1.0) - f32(2.0) * nu)))  # physics.rs:29-31
        self.hardening_enabled, self.hardening_factor, self.cohesion = bool(hardening_enabled), hardening_factor, cohesion
        if m is not None:
            self.friction = m
        else:
            sin_f, d = np.sin(f32(friction_angle), dtype=f32), f32(3.0)
            self.friction = float(np.sqrt(f32(2.0) / f32(3.0), dtype=f32) * f32(2.0) * sin_f / (f32(3.0) - sin_f) * d / np.sqrt(f32(2.0) / (f32(6.0) - d), dtype=f32))

    def to_core_model(self):
        m = abi.PlasticModelC()
        m.tag = abi.PLASTIC_NACC
        n = m.u.nacc
        n.mu, n.kappa, n.hardening_enabled = self.mu, self.kappa, int(self.hardening_enabled)
        n.hardening_factor, n.cohesion, n.friction = self.hardening_factor, self.cohesion, self.friction
        return m


class MaximumStressFailure:
    """`MaximumStressFailure::new(max_principal_stress, max_shear_stress)` (failure_maximum_stress.rs:13-18)."""

    def __init__(self, max_principal_stress, max_shear_stress):
        self.max_principal_stress, self.max_shear_stress = max_principal_stress, max_shear_stress

    def to_core_model(self):
        m = abi.FailureModelC()
        m.tag = abi.FAILURE_MAXIMUM_STRESS
        m.u.maximum_stress.max_principal_stress, m.u.maximum_stress.max_shear_stress = self.max_principal_stress, self.max_shear_stress
        return m


class ParticleModel:
    """`ParticleModel::{new, with_plasticity}` (src/dynamics/particle_model.rs:85-113)."""

    def __init__(self, constitutive_model, plastic_model=None, failure_model=None):
        self.constitutive_model = constitutive_model
        self.plastic_model = plastic_model
        self.failure_model = failure_model

    @classmethod
    def new(cls, constitutive_model):
        return cls(constitutive_model)

    @classmethod
    def with_plasticity(cls, constitutive_model, plasticity):
        return cls(constitutive_model, plastic_model=plasticity)

    @classmethod
    def with_failure(cls, constitutive_model, failure):     # particle_model.rs `with_failure`
        return cls(constitutive_model, failure_model=failure)

    def to_c(self):
        m = abi.ParticleModelC()
        m.constitutive_model = self.constitutive_model.to_core_model()
        if self.plastic_model is not None:
            m.has_plastic_model = 1
            m.plastic_model = self.plastic_model.to_core_model()
        if self.failure_model is not None:
            m.has_failure_model = 1
            m.failure_model = self.failure_model.to_core_model()
        return m


class ParticleModelSet:
    """`ParticleModelSet::{new, insert, get, iter}` (particle_model.rs:117-139); handles are indices."""

    def __init__(self):
        self.models = []

    def insert(self, model):
        self.models.append(model)
        return len(self.models) - 1

    def get(self, handle):
        return self.models[handle]

    def __iter__(self):
        return iter(enumerate(self.models))

    def __len__(self):
        return len(self.models)

    def to_c_array(self):
        arr = (abi.ParticleModelC * max(len(self.models), 1))()
        for i, m in enumerate(self.models):
            arr[i] = m.to_c()
        return arr
