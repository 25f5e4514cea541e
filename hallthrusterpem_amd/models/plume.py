"""Plume model -- GPU drop-in for `hallmd.models.plume` (src/hallmd/models/plume.py)."""
import ctypes as C

import numpy as np

from .. import _lib, _marshal as m, constants

__all__ = ['current_density', 'angle_grid']

_KEYS = ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex', 'I_B0')


def angle_grid() -> np.ndarray:
    """The fixed sweep of plume.py:53, np.linspace(0, pi/2, 91), as the library holds it."""
    return np.ctypeslib.as_array(_lib.load().pem_angle_grid(), shape=(_lib.NANGLE,)).copy()


def _coords(shape, grid):
    """plume.py:152-157: an object array over the loop shape, every cell the same 91-vector.

    The reference fills it in a Python loop over all N samples (1-4 % of its run time).  Up to 4096 cells this
    does the same (a writable array, as the reference returns); beyond that it returns a zero-stride, read-only
    view of a single cell -- O(1) instead of ~20 ns per sample, same values under indexing and iteration."""
    n = int(np.prod(shape))
    if n <= 4096:
        return np.frompyfunc(lambda _: grid, 1, 1)(np.empty(shape, dtype=np.uint8))
    cell = np.empty((), dtype=object)
    cell[()] = grid
    return np.broadcast_to(cell, shape)


def current_density(inputs: dict, sweep_radius=1.0) -> dict:
    """Semi-empirical ion current density over a 0..90 degree sweep; mirrors plume.py:21-159.

    :param inputs: `P_b` (Torr), `c0`..`c5`, `sigma_cex` (m^2), `I_B0` (A); optional `T` (N).
    :param sweep_radius: radius or radii (m) of the sweep; more than one adds a trailing axis.
    :returns: `j_ion` (..., 91[, R]), `div_angle` (...[, R]), `T_c` if `T` was given, and `j_ion_coords`
              (object array over the loop shape, each cell the 91 angles in rad) -- shapes, squeeze rule
              and the 1e-20 fill of invalid samples exactly as the reference.
    """
    vals = [inputs[k] for k in _KEYS]
    thrust = inputs.get('T', None)
    radii = np.atleast_1d(np.asarray(sweep_radius, dtype=np.float64)).reshape(-1)
    R = int(radii.size)
    lib = _lib.load()
    shape = m.loop_shape(vals + ([thrust] if thrust is not None else []))
    n = int(np.prod(shape))
    grid = angle_grid()
    radii = np.ascontiguousarray(radii)
    tail = (R,) if R > 1 else ()

    if m.any_device_tensor(vals + ([thrust] if thrust is not None else [])):
        import torch
        dev = m.pick_device(vals + ([thrust] if thrust is not None else []))
        with torch.cuda.device(dev):
            flat = [m.dev_flat(v, shape, dev) for v in vals]
            t_in = m.dev_flat(thrust, shape, dev) if thrust is not None else None
            j = torch.empty(n * _lib.NANGLE * R, dtype=torch.float64, device=dev)
            div = torch.empty(n * R, dtype=torch.float64, device=dev)
            tc = torch.empty(n * R, dtype=torch.float64, device=dev) if thrust is not None else None
            _lib.check(lib.pem_plume_f64_dev(n, R, m.np_ptr(radii), constants.TORR_2_PA, *[m.t_ptr(t) for t in flat],
                                             m.t_ptr(t_in), m.t_ptr(j), m.t_ptr(div), m.t_ptr(tc), None,
                                             m.current_stream_ptr(dev)))
        ret = {'j_ion': j.reshape(shape + (_lib.NANGLE,) + tail), 'div_angle': div.reshape(shape + tail)}
        if thrust is not None:
            ret['T_c'] = tc.reshape(shape + tail)
        ret['j_ion_coords'] = _coords(shape, grid)
        return ret

    flat = [m.host_flat(v, shape) for v in vals]
    t_in = m.host_flat(thrust, shape) if thrust is not None else None
    j = m.host_empty(n * _lib.NANGLE * R)
    div = np.empty(n * R, dtype=np.float64)
    tc = np.empty(n * R, dtype=np.float64) if thrust is not None else None
    _lib.check(lib.pem_plume_f64(n, R, m.np_ptr(radii), constants.TORR_2_PA, *[m.np_ptr(a) for a in flat],
                                 m.np_ptr(t_in), m.np_ptr(j), m.np_ptr(div), m.np_ptr(tc), None))
    ret = {'j_ion': j.reshape(shape + (_lib.NANGLE,) + tail), 'div_angle': div.reshape(shape + tail)}
    if thrust is not None:
        ret['T_c'] = tc.reshape(shape + tail)
    ret['j_ion_coords'] = _coords(shape, grid)
    return ret
