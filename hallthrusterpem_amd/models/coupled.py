"""Coupled PEM-v0 forward pass cathode -> thruster -> plume in ONE kernel launch.

The graph is the one scripts/pem_v0/pem_v0_SPT-100.yml declares (Cathode `V_cc` -> Thruster; Thruster
`I_B0` -> Plume).  The thruster stage is the reference's analytic test double
(tests/sim_hallthruster.jl:35-48, see models/thruster.py); with it the three stages fuse into a single
streaming map of 15 inputs -> (V_cc, I_B0, T, j_ion[91], div_angle, T_c) per Monte-Carlo sample.
"""
import numpy as np

from .. import _lib, _marshal as m, constants
from .plume import _coords, angle_grid

__all__ = ['pem_v0_coupled', 'COUPLED_INPUTS']

COUPLED_INPUTS = ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T', 'mdot_a', 'a_1',
                  'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')


def pem_v0_coupled(inputs: dict, sweep_radius: float = 1.0, profile: bool = True, coords: bool = False,
                   out: dict | None = None) -> dict:
    """Evaluate the coupled model for every sample.

    :param inputs: the 15 arrays of `COUPLED_INPUTS` (scalars broadcast).  numpy -> numpy; CUDA tensors -> CUDA
                   tensors on the current stream, without a host round trip.
    :param sweep_radius: the single plume sweep radius in metres (pem_v0_SPT-100.yml:218 uses 1.0).
    :param profile: False selects the reduced-QoI mode: `j_ion` is neither computed into memory nor returned.
    :param coords: also return `j_ion_coords` (an object array like plume.current_density's).
    :param out: device path only -- preallocated flat CUDA tensors to write into (keys of the result), so a
                sampling loop can reuse its buffers.
    :returns: `V_cc`, `I_B0`, `T`, `div_angle`, `T_c`, `invalid` (bool; plume.py:105) and, if `profile`, `j_ion`.
    """
    vals = [inputs[k] for k in COUPLED_INPUTS]
    lib = _lib.load()
    shape = m.loop_shape(vals)
    n = int(np.prod(shape))
    radius = float(sweep_radius)
    names = ['V_cc', 'I_B0', 'T', 'div_angle', 'T_c']

    if m.any_device_tensor(vals):
        import torch
        dev = m.pick_device(vals)
        with torch.cuda.device(dev):
            flat = [m.dev_flat(v, shape, dev) for v in vals]
            o = dict(out) if out else {}
            for k in names:
                if k not in o:
                    o[k] = torch.empty(n, dtype=torch.float64, device=dev)
            if profile and 'j_ion' not in o:
                o['j_ion'] = torch.empty(n * _lib.NANGLE, dtype=torch.float64, device=dev)
            if 'invalid' not in o:
                o['invalid'] = torch.empty(n, dtype=torch.uint8, device=dev)
            _lib.check(lib.pem_coupled_f64_dev(
                n, constants.TORR_2_PA, radius, *[m.t_ptr(t) for t in flat], m.t_ptr(o['V_cc']), m.t_ptr(o['I_B0']),
                m.t_ptr(o['T']), m.t_ptr(o['j_ion']) if profile else None, m.t_ptr(o['div_angle']), m.t_ptr(o['T_c']),
                m.t_ptr(o['invalid']), m.current_stream_ptr(dev)))
        ret = {k: o[k].reshape(shape) for k in names}
        ret['invalid'] = o['invalid'].reshape(shape).bool()
        if profile:
            ret['j_ion'] = o['j_ion'].reshape(shape + (_lib.NANGLE,))
    else:
        flat = [m.host_flat(v, shape) for v in vals]
        o = {k: np.empty(n, dtype=np.float64) for k in names}
        j = m.host_empty(n * _lib.NANGLE) if profile else None
        inv = np.zeros(n, dtype=np.uint8)
        _lib.check(lib.pem_coupled_f64(n, constants.TORR_2_PA, radius, *[m.np_ptr(a) for a in flat],
                                       m.np_ptr(o['V_cc']), m.np_ptr(o['I_B0']), m.np_ptr(o['T']), m.np_ptr(j),
                                       m.np_ptr(o['div_angle']), m.np_ptr(o['T_c']), m.np_ptr(inv)))
        ret = {k: o[k].reshape(shape) for k in names}
        ret['invalid'] = inv.reshape(shape).astype(bool)
        if profile:
            ret['j_ion'] = j.reshape(shape + (_lib.NANGLE,))
    if coords and profile:
        ret['j_ion_coords'] = _coords(shape, angle_grid())
    return ret
