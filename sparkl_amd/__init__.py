"""sparkl_amd -- MI355X-native MPM hot path behind sparkl's pipeline API."""
from . import abi, models, particles, scenes  # noqa: F401
from .abi import SolverParameters  # noqa: F401
from .models import (CorotatedLinearElasticity, DruckerPragerPlasticity, MonaghanSphEos, NeoHookeanElasticity,  # noqa: F401
                     ParticleModel, ParticleModelSet)
from .particles import ParticleComponents, ParticleSet, extract_particles_components, new_particles  # noqa: F401


def __getattr__(name):
    if name == "MpmPipeline":
        from .pipeline import MpmPipeline
        return MpmPipeline
    raise AttributeError(name)
