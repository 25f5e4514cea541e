"""MI355X-native batched evaluator for the PEM-v0 cathode -> thruster -> plume sub-models.

Drop-in for the vectorised model callables of `hallmd.models` (JANUS-Institute/HallThrusterPEM):

    from hallthrusterpem_amd.models import cathode_coupling, current_density

The arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of include/pem_hip.h
(libpem_hip.so, loaded with ctypes).  There is no CPU implementation in this package.
"""
__version__ = '0.1.0'

from . import constants  # noqa: F401


def set_device(index: int):
    """Select this process's GPU for every thread (see `_lib.set_device`)."""
    from . import _lib
    _lib.set_device(index)
