"""Model callables with the names, signatures and return dictionaries of `hallmd.models`
(src/hallmd/models/__init__.py:15-19 of the reference), evaluated on the GPU.

An amisc YAML can point at them unchanged apart from the package name, e.g.
`model: !!python/name:hallthrusterpem_amd.models.cathode.cathode_coupling` (cf. pem_v0_SPT-100.yml:6).
"""
from .cathode import cathode_coupling
from .coupled import pem_v0_coupled
from .plume import current_density
from .thruster import PEM_TO_JULIA, hallthruster_jl, thruster_analytic

# the reference's three (src/hallmd/models/__init__.py:15-19) first, then what this package adds
__all__ = ['cathode_coupling', 'hallthruster_jl', 'current_density', 'thruster_analytic', 'pem_v0_coupled', 'PEM_TO_JULIA']
