"""Cathode coupling model -- GPU drop-in for `hallmd.models.cathode` (src/hallmd/models/cathode.py)."""
import numpy as np

from .. import _lib, _marshal as m, constants

__all__ = ['cathode_coupling']

_KEYS = ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')


def cathode_coupling(inputs: dict) -> dict:
    """Cathode coupling voltage vs background pressure (Jorns 2021); mirrors cathode.py:16-38.

    :param inputs: `P_b` (Torr), `V_a` (V), `T_e` (eV), `V_vac` (V), `Pstar` (Torr), `P_T` (Torr); scalars or
                   arrays that broadcast together.  numpy in -> numpy out; CUDA torch tensors in -> CUDA
                   tensors out (no host round trip).
    :returns: `{'V_cc': float64 array}`, always at least 1-D (cathode.py:34).

    Deviation from the reference, on purpose: a scalar `V_a` with array inputs is broadcast; the reference
    raises IndexError there whenever a clipped index is > 0 (cathode.py:37).
    A missing key raises KeyError as dict indexing does in the reference.
    """
    vals = [inputs[k] for k in _KEYS]
    lib = _lib.load()
    shape = m.loop_shape(vals)
    n = int(np.prod(shape))
    if m.any_device_tensor(vals):
        import torch
        dev = m.pick_device(vals)
        with torch.cuda.device(dev):
            flat = [m.dev_flat(v, shape, dev) for v in vals]
            out = torch.empty(n, dtype=torch.float64, device=dev)
            _lib.check(lib.pem_cathode_f64_dev(n, *[m.t_ptr(t) for t in flat], constants.TORR_2_PA, m.t_ptr(out),
                                               m.current_stream_ptr(dev)))
        return {'V_cc': out.reshape(shape)}
    flat = [m.host_flat(v, shape) for v in vals]
    out = np.empty(n, dtype=np.float64)
    _lib.check(lib.pem_cathode_f64(n, *[m.np_ptr(a) for a in flat], constants.TORR_2_PA, m.np_ptr(out)))
    return {'V_cc': out.reshape(shape)}
