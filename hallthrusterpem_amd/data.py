"""`hallmd.data.pem_to_xarray` (src/hallmd/data.py:239-279): model outputs in the layout the calibration data are compared in.

The reference builds `xarray.DataArray`s inside `pem_core.data.DataEntry / DataField` records; neither package is in this image
(SURVEY.md section 8c), so both are imported lazily: with them the function returns the reference's own structure, without
them the same nested records as plain dicts whose array leaves are `{'val': ndarray, 'coords': {...}, 'dims': (...)}`.
Parity UNPINNED: the reference holds no test or fixture for this function and it cannot be imported here; what is restated is
its indexing -- the LAST sweep radius' corrected thrust, `j_ion` transposed to (r, theta), `u_ion` on its `z` coordinates.
The rest of `hallmd.data` (CSV loading of thruster measurements through pem_core) is out of scope.
"""
import numpy as np


def _host(x):
    """numpy view of a model output: device tensors are copied to the host, object arrays and lists are left to numpy"""
    if hasattr(x, 'detach') and hasattr(x, 'cpu'):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _of_sample(coords, i, length):
    """The coordinate vector of sample i: the reference indexes per-sample coordinates (`outputs['j_ion_coords'][i]`, an object
    array whose cells are the grid, plume.py:152-157); a single grid shared by every sample -- what the batched thruster stage
    of this package returns for `u_ion_coords` -- is accepted as it is."""
    c = coords if isinstance(coords, np.ndarray) and coords.dtype == object else _host(coords)
    if c.dtype != object and c.ndim == 1 and c.shape[0] == length:
        return np.asarray(c, dtype=np.float64)
    return np.asarray(c[i], dtype=np.float64)


def _backend():
    try:
        import xarray as xr
        from pem_core.data import DataEntry, DataField
        return xr, DataEntry, DataField
    except ImportError:
        return None, None, None


def pem_to_xarray(operating_conditions: list, outputs: dict, sweep_radii, use_corrected_thrust: bool = True) -> list:
    """Convert the outputs of the Hall thruster PEM into one data entry per operating condition (src/hallmd/data.py:239-279).

    :param operating_conditions: one dict per sample i of `outputs` (flow rate, pressure, voltage, ...), passed through.
    :param outputs: model outputs over N samples: `T_c` (N,) or (N, R) [or `T` (N,)], `I_d` (N,), `V_cc` (N,), `u_ion` (N, nz)
                    with `u_ion_coords`, `j_ion` (N, 91) or (N, 91, R) with `j_ion_coords`; numpy arrays or CUDA tensors.
    :param sweep_radii: the R radii of `j_ion`'s last axis, sorted (a length-1 array when the radius axis was squeezed).
    :param use_corrected_thrust: thrust = `T_c` at the LAST radius (data.py:250-252) instead of the uncorrected `T`.
    :returns: a list of entries `{operating_condition, data}` with data fields "discharge current" (A), "cathode coupling
              voltage" (V), "thrust" (N), "ion velocity" (m/s; dims ("z",)) and "ion current density" (A/m^2; dims ("r", "theta")):
              `pem_core.data.DataEntry` objects holding `xarray.DataArray`s when both packages are importable, else dicts with
              `{'val': {'val': ndarray, 'coords': {dim: ndarray}, 'dims': tuple}, 'unit': str}` fields.
    """
    xr, DataEntry, DataField = _backend()
    r = np.atleast_1d(np.asarray(_host(sweep_radii), dtype=np.float64))
    T_c = _host(outputs['T_c']) if use_corrected_thrust else None
    T = None if use_corrected_thrust else _host(outputs['T'])
    I_d, V_cc = _host(outputs['I_d']), _host(outputs['V_cc'])
    u_ion = _host(outputs['u_ion'])
    j_ion = np.atleast_3d(_host(outputs['j_ion']))                       # (N, 91) -> (N, 91, 1): data.py:263
    if j_ion.shape[2] != r.size:
        raise ValueError(f'j_ion has {j_ion.shape[2]} radii but {r.size} sweep radii were given')

    def array(val, coords=None, dims=()):
        val = np.asarray(val, dtype=np.float64)
        if xr is not None:
            return xr.DataArray(val, coords=[coords[d] for d in dims], dims=list(dims)) if dims else xr.DataArray(val)
        return {'val': val, 'coords': dict(coords or {}), 'dims': tuple(dims)}

    def field(val, unit):
        return DataField(val=val, unit=unit) if DataField is not None else {'val': val, 'unit': unit}

    entries = []
    for i, opcond in enumerate(operating_conditions):
        # with several radii there are several corrected thrusts: the last one, the radii being sorted (data.py:250-252)
        thrust = np.atleast_1d(T_c[i])[-1] if use_corrected_thrust else T[i]
        z = _of_sample(outputs['u_ion_coords'], i, u_ion.shape[-1])
        theta = _of_sample(outputs['j_ion_coords'], i, j_ion.shape[1])
        instance = {
            'discharge current': field(array(I_d[i]), 'A'),
            'cathode coupling voltage': field(array(V_cc[i]), 'V'),
            'thrust': field(array(thrust), 'N'),
            'ion velocity': field(array(u_ion[i], {'z': z}, ('z',)), 'm/s'),
            'ion current density': field(array(j_ion[i, :, :].T, {'r': r, 'theta': theta}, ('r', 'theta')), 'A/m^2'),
        }
        entries.append(DataEntry(operating_condition=opcond, data=instance) if DataEntry is not None
                       else {'operating_condition': opcond, 'data': instance})
    return entries
